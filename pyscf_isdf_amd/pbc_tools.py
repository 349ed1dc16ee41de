"""Host-side helpers with the reference's conventions (small, O(G) work done once per q):
the Coulomb kernel for a difference vector q (pyscf/pbc/tools/pbc.py:230-420, exxdiv=None branch,
including the wrap-around of k+G beyond the mesh edge :272-302 and the zeroing of edge components
:400-401) and the set of distinct difference vectors of a k-point list."""
import numpy as np


def get_coulG(cell, q=np.zeros(3), mesh=None, wrap_around=True, omega=None):
    if mesh is None:
        mesh = cell.mesh
    a = np.asarray(cell.lattice_vectors(), dtype=float)
    b = 2 * np.pi * np.linalg.inv(a.T)
    rx = np.fft.fftfreq(mesh[0], 1. / mesh[0])
    ry = np.fft.fftfreq(mesh[1], 1. / mesh[1])
    rz = np.fft.fftfreq(mesh[2], 1. / mesh[2])
    Gv = (rx[:, None, None, None] * b[0] + ry[None, :, None, None] * b[1] + rz[None, None, :, None] * b[2]).reshape(-1, 3)
    q = np.asarray(q, dtype=float)
    nonzero = abs(q).sum() > 1e-9
    kG = q + Gv if nonzero else Gv
    equal2boundary = None
    if wrap_around and nonzero:
        equal2boundary = np.zeros(Gv.shape[0], dtype=bool)
        box_edge = np.einsum('i,ij->ij', np.asarray(mesh) // 2 + 0.5, b)
        if not all(np.linalg.solve(box_edge.T, q).round(9).astype(int) == 0):
            raise ValueError('q lies outside the first FFT box (pbc.py:281)')
        reduced = np.linalg.solve(box_edge.T, kG.T).T.round(9)
        on_edge = reduced.astype(int)
        for x in range(3):
            equal2boundary |= reduced[:, x] == 1
            equal2boundary |= reduced[:, x] == -1
            kG[on_edge[:, x] == 1] -= 2 * box_edge[x]
            kG[on_edge[:, x] == -1] += 2 * box_edge[x]
    absG2 = np.einsum('gi,gi->g', kG, kG)
    with np.errstate(divide='ignore'):
        coulG = 4 * np.pi / absG2
    coulG[absG2 == 0] = 0
    if equal2boundary is not None:
        coulG[equal2boundary] = 0
    if omega:                                   # range separation, pyscf/pbc/tools/pbc.py:408-418
        e = np.exp(-.25 / omega ** 2 * absG2)
        coulG = coulG * (e if omega > 0 else 1 - e)
    return coulG


def unique_q(kpts, kpts_band=None, tol=1e-9):
    """Distinct q = k2 - k1 (k1 in band, k2 in kpts) and index[k1][k2] -> position in the q list."""
    kpts = np.reshape(kpts, (-1, 3))
    band = kpts if kpts_band is None else np.reshape(kpts_band, (-1, 3))
    qs, index = [], np.zeros((len(band), len(kpts)), dtype=int)
    for i1, k1 in enumerate(band):
        for i2, k2 in enumerate(kpts):
            q = k2 - k1
            for iq, qq in enumerate(qs):
                if abs(qq - q).max() < tol:
                    index[i1, i2] = iq
                    break
            else:
                qs.append(q)
                index[i1, i2] = len(qs) - 1
    return np.array(qs), index

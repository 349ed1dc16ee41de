"""Host-side helper of the k-point path: the set of distinct difference vectors q = k2 - k1 of a k-point list.
(The Coulomb kernel table for a q lives on the device: include/mi355_isdf.h isdf_coulG_q.)"""
import numpy as np


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


def wigner_seitz_kernel(a, nk):
    """The tabulated part of exxdiv='vcut_ws' (PRB 87, 165122; the role of pyscf/pbc/tools/pbc.py:422-480).  For a k-mesh of
    nk = (n1, n2, n3) points the exchange hole lives in the nk-fold cell with lattice A = diag(nk) a (ROWS scaled: a_i -> nk_i a_i.
    Deliberate deviation: the reference writes `cell.lattice_vectors() * Nk` (pbc.py:434), which scales COLUMNS; the two agree
    for orthogonal cells and isotropic k-meshes - the only case the reference pins - and differ for an anisotropic mesh on a
    non-orthogonal cell, which is parity unpinned).  The Coulomb kernel is
    split as erfc(alpha r)/r + erf(alpha r)/r; the first part is short-ranged and keeps its analytic transform, the second is
    cut at the Wigner-Seitz cell of A: it is tabulated on a real-space mesh of A as erf(alpha r_min)/r_min with r_min the
    distance to the nearest lattice point (the corners of the parallelepiped, by periodicity) and Fourier transformed once.
    alpha = 5 / R_in with R_in half the smallest spacing of A's lattice planes, so erfc has decayed to ~1e-11 at the cell
    surface; the mesh has 4 int(3 alpha L_i) points along axis i.  Returns dict(alpha, a, mesh, vq, maxq) for
    isdf_set_coulomb_ws: vq on the reciprocal lattice of A in fftfreq order, maxq the largest |component| it covers."""
    import scipy.special
    A = np.asarray(a, dtype=float) * np.asarray(nk, dtype=float).reshape(3, 1)
    spacing = 1.0 / np.linalg.norm(np.linalg.inv(A), axis=0)
    alpha = 5.0 / (spacing.min() / 2.0)
    mesh = np.array([4 * int(L * alpha * 3.0) for L in spacing])       # the reference's operation order: the mesh must be the same
    n_tot = int(np.prod(mesh))
    # distance of every mesh point to its nearest lattice point: fold the fractional coordinate about each of the 8 corners
    f = [np.arange(n) / n for n in mesh]
    r2min = np.full(tuple(mesh), np.inf)
    for c0 in (0.0, 1.0):
        for c1 in (0.0, 1.0):
            for c2 in (0.0, 1.0):
                d = ((f[0] - c0)[:, None, None, None] * A[0] + (f[1] - c1)[None, :, None, None] * A[1]
                     + (f[2] - c2)[None, None, :, None] * A[2])
                r2min = np.minimum(r2min, np.einsum('xyzc,xyzc->xyz', d, d))
    r = np.sqrt(r2min)
    with np.errstate(divide='ignore', invalid='ignore'):
        v = np.where(r > 1e-9, scipy.special.erf(alpha * r) / r, 2.0 * alpha / np.sqrt(np.pi))
    vq = np.fft.fftn(v) * (abs(np.linalg.det(A)) / n_tot)
    if abs(vq.imag).max() > 1e-6:
        raise RuntimeError('Unconventional lattice was found')       # the reference's own diagnosis (pbc.py:466-473)
    B = 2 * np.pi * np.linalg.inv(A.T)
    freqs = [np.fft.fftfreq(n, 1.0 / n) for n in mesh]
    Gk = (freqs[0][:, None, None, None] * B[0] + freqs[1][None, :, None, None] * B[1] + freqs[2][None, None, :, None] * B[2])
    maxq = abs(Gk.reshape(-1, 3)).max(axis=0)
    return dict(alpha=float(alpha), a=A, mesh=mesh, vq=np.ascontiguousarray(vq.real).ravel(), maxq=maxq)

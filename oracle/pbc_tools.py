"""Oracle: FFT conventions, Coulomb kernel, fingerprint.  TEST INFRASTRUCTURE ONLY.

Follows pyscf/pbc/tools/pbc.py:149-211 (fft unscaled forward, ifft scaled by 1/G, batch over the
leading axis, C order over the mesh), :230-420 (get_coulG: 4 pi/|k+G|^2 with G=0 -> 0, wrap-around of
k+G beyond the mesh edge and zeroing of edge components for k != 0), :483-511 (madelung, not
restated: parity runs use exxdiv=None like pyscf/pbc/df/test/test_fft.py:582,618,632,643) and
pyscf/lib/misc.py:1150-1154 (fp).
"""
import numpy as np
import scipy.fft

WORKERS = -1


def fft(f, mesh):
    f = np.asarray(f)
    if f.size == 0:
        return np.zeros_like(f)
    f3d = f.reshape(-1, *mesh)
    g3d = scipy.fft.fftn(f3d, axes=(1, 2, 3), workers=WORKERS)
    ngrids = np.prod(mesh)
    if f.ndim == 1 or (f.ndim == 3 and f.size == ngrids):
        return g3d.ravel()
    return g3d.reshape(-1, ngrids)


def ifft(g, mesh):
    g = np.asarray(g)
    if g.size == 0:
        return np.zeros_like(g)
    g3d = g.reshape(-1, *mesh)
    f3d = scipy.fft.ifftn(g3d, axes=(1, 2, 3), workers=WORKERS)
    ngrids = np.prod(mesh)
    if g.ndim == 1 or (g.ndim == 3 and g.size == ngrids):
        return f3d.ravel()
    return f3d.reshape(-1, ngrids)


def get_Gv(b, mesh):
    rx = np.fft.fftfreq(mesh[0], 1. / mesh[0])
    ry = np.fft.fftfreq(mesh[1], 1. / mesh[1])
    rz = np.fft.fftfreq(mesh[2], 1. / mesh[2])
    Gv = (rx[:, None, None, None] * b[0] + ry[None, :, None, None] * b[1] + rz[None, None, :, None] * b[2])
    return Gv.reshape(-1, 3)


def get_coulG(a, mesh, k=np.zeros(3), wrap_around=True, omega=None, rc=None):
    """Coulomb kernel on the FFT mesh for lattice ``a`` (3,3 Bohr); exxdiv=None semantics, or - rc given - the spherically
    truncated kernel of exxdiv='vcut_sph' (pbc.py:312-317) with Rc = rc."""
    a = np.asarray(a, dtype=float)
    b = 2 * np.pi * np.linalg.inv(a.T)
    Gv = get_Gv(b, mesh)
    k = np.asarray(k, dtype=float)
    if abs(k).sum() > 1e-9:
        kG = k + Gv
    else:
        kG = Gv
    equal2boundary = None
    if wrap_around and abs(k).sum() > 1e-9:
        equal2boundary = np.zeros(Gv.shape[0], dtype=bool)
        box_edge = np.einsum('i,ij->ij', np.asarray(mesh) // 2 + 0.5, b)
        assert all(np.linalg.solve(box_edge.T, k).round(9).astype(int) == 0)
        reduced = np.linalg.solve(box_edge.T, kG.T).T.round(9)
        on_edge = reduced.astype(int)
        for x in range(3):
            equal2boundary |= reduced[:, x] == 1
            equal2boundary |= reduced[:, x] == -1
            kG[on_edge[:, x] == 1] -= 2 * box_edge[x]
            kG[on_edge[:, x] == -1] += 2 * box_edge[x]
    absG2 = np.einsum('gi,gi->g', kG, kG)
    with np.errstate(divide='ignore', invalid='ignore'):
        coulG = 4 * np.pi / absG2
        if rc:
            coulG = coulG * (1.0 - np.cos(np.sqrt(absG2) * rc))
    coulG[absG2 == 0] = 4 * np.pi * 0.5 * rc ** 2 if rc else 0
    if equal2boundary is not None:
        coulG[equal2boundary] = 0
    if omega:                                   # range separation, pyscf/pbc/tools/pbc.py:408-418
        e = np.exp(-.25 / omega ** 2 * absG2)
        coulG = coulG * (e if omega > 0 else 1 - e)
    return coulG


def fp(a):
    a = np.asarray(a)
    return np.dot(np.cos(np.arange(a.size)), a.ravel())

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


def get_coulG(a, mesh, k=np.zeros(3), wrap_around=True, omega=None, rc=None, ws=None):
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
        if ws is not None:
            coulG = coulG * (1.0 - np.exp(-absG2 / (4 * ws['alpha'] ** 2)))
    coulG[absG2 == 0] = 4 * np.pi * 0.5 * rc ** 2 if rc else 0
    if ws is not None:
        # exxdiv='vcut_ws' (pbc.py:318-346): the short-range part above (its G -> 0 limit is pi / alpha^2) plus the precomputed
        # transform of erf(alpha r)/r truncated to the Wigner-Seitz cell of the nk-fold lattice, looked up at k + G
        coulG[absG2 == 0] = np.pi / ws['alpha'] ** 2
        gxyz = np.dot(kG, ws['a'].T) / (2 * np.pi)
        gxyz = gxyz.round(decimals=6).astype(int)
        kmesh = np.asarray(ws['mesh'])
        gxyz = (gxyz + kmesh) % kmesh
        qidx = (gxyz[:, 0] * kmesh[1] + gxyz[:, 1]) * kmesh[2] + gxyz[:, 2]
        inside = (abs(kG) <= ws['maxq']).all(axis=1)
        coulG[inside] += ws['vq'][qidx[inside]]
    if equal2boundary is not None:
        coulG[equal2boundary] = 0
    if omega:                                   # range separation, pyscf/pbc/tools/pbc.py:408-418
        e = np.exp(-.25 / omega ** 2 * absG2)
        coulG = coulG * (e if omega > 0 else 1 - e)
    return coulG


def precompute_exx(a, nk):
    """The Wigner-Seitz truncated kernel table of exxdiv='vcut_ws' (pbc.py:422-480, PRB 87 165122): on the lattice a * nk
    (the cell the k-mesh stands for) tabulate erf(alpha r)/r with r the distance to the NEAREST lattice corner, alpha =
    5 / (half the smallest plane spacing), on a mesh of 4 int(3 alpha L_i) points per axis, and transform it."""
    import scipy.special
    ak = np.asarray(a, dtype=float) * np.asarray(nk, dtype=float)[:, None]
    Lc = 1.0 / np.linalg.norm(np.linalg.inv(ak), axis=0)
    alpha = 5.0 / (Lc.min() / 2.0)
    mesh = np.array([4 * int(L * alpha * 3.0) for L in Lc])
    frac = np.stack(np.meshgrid(*[np.arange(n) / n for n in mesh], indexing='ij'), axis=-1).reshape(-1, 3)
    rs = frac.dot(ak)
    corners = np.array([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)], dtype=float).dot(ak)
    r = np.min([np.linalg.norm(rs - c, axis=1) for c in corners], axis=0)
    vR = scipy.special.erf(alpha * r) / (r + 1e-200)
    vR[r < 1e-9] = 2 * alpha / np.sqrt(np.pi)
    vol = abs(np.linalg.det(ak))
    vG = (vol / len(rs)) * fft(vR, mesh)
    if abs(vG.imag).max() > 1e-6:
        raise RuntimeError('Unconventional lattice was found')
    bk = 2 * np.pi * np.linalg.inv(ak.T)
    return dict(alpha=alpha, a=ak, mesh=mesh, vq=vG.real.copy(), maxq=abs(get_Gv(bk, mesh)).max(axis=0))


def fp(a):
    a = np.asarray(a)
    return np.dot(np.cos(np.arange(a.size)), a.ravel())

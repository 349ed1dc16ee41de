"""Debug aid (test infrastructure): block-Jacobi route stage by stage, HIP backend vs the checker backend."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
import test_gpu_kpts as T
from oracle_backend import OracleBackend
from pyscf_isdf_amd.backend import HipBackend

cell, coords, Ls, rcut, kpts, aos, dms = T._setup()
nk, nao, G = len(kpts), cell.nao_nr(), len(coords)
nh = nk * nao
hb, ob = HipBackend(0), OracleBackend()
X = T._gpu_X(hb, cell, coords, Ls, rcut, kpts)
Xh = hb.to_host(X)
Xo = torch.from_numpy(Xh.copy())
from pyscf_isdf_amd.isdf import ISDF
df = ISDF(cell, kpts=kpts, c_isdf=int(sys.argv[1]) if len(sys.argv) > 1 else 10, select='local', backend=ob)
df.fit_route = 'cholesky'
df.build()
ip = df.ip
P = len(ip)
ip_off = np.array([0, P // 2, P], dtype=np.int32)
print('P', P)

def rel(a, b):
    return abs(a - b).max() / abs(b).max()

out = {}
for name, be, Xb in (('hip', hb, X), ('cpu', ob, Xo)):
    d_ip = be.to_device(ip)
    aoP = be.empty((P, 2 * nh))
    be.gather_aoP(Xb, d_ip, aoP)
    A = be.empty((P, P)); be.gram_sq(aoP, A, nh)
    A0 = be.to_host(A)
    D = be.empty((P, P)); be.block_chol(A, ip_off, 0.0, D)
    be.block_solve(D, ip_off, 0, 0, A); be.block_solve(D, ip_off, 1, 1, A)
    Ap = be.to_host(A)
    reg = be.chol_inplace(A, 1e-12)
    B = be.empty((P, G)); be.pair_gram_rows(aoP, Xb, G, B, nh)
    B0 = be.to_host(B)
    be.block_solve(D, ip_off, 0, 0, B)
    out[name] = dict(aoP=be.to_host(aoP), A=A0, D=be.to_host(D), Ap=Ap, U=np.triu(be.to_host(A).T).T if False else be.to_host(A), B=B0, Y=be.to_host(B), reg=reg)
for k in ('aoP', 'A', 'D', 'Ap', 'B', 'Y'):
    print(k, rel(out['hip'][k], out['cpu'][k]))
print('reg', out['hip']['reg'], out['cpu']['reg'])
w = np.linalg.eigvalsh(out['cpu']['Ap'])
print('cond A prime', w[-1] / w[0], 'min', w[0], 'cond A', np.linalg.cond(out['cpu']['A']))
L = np.tril(out['cpu']['U']); Lh = np.tril(out['hip']['U'])
print('chol factor', rel(Lh, L))

# ---- M'^q and the finishing steps
from pyscf_isdf_amd import pbc_tools
mesh = np.asarray(cell.mesh, dtype=np.int32)
w = cell.vol / G
qs, qidx = pbc_tools.unique_q(kpts)
for iq, q in enumerate(qs):
    res = {}
    for name, be in (('hip', hb), ('cpu', ob)):
        Y = be.to_device(out[name]['Y'])
        U = be.to_device(out[name]['U'])
        D = be.to_device(out[name]['D'])
        coulG = be.to_device(pbc_tools.get_coulG(cell, q, mesh))
        Wre = be.empty((P, P)); Wim = be.empty((P, P))
        be.coulomb_Wq(Y, mesh, coulG, w, 0, P, 64, Wre, Wim, upper_only=True)
        be.symmetrize_hermitian(Wre, Wim)
        M = be.to_host(Wre) + 1j * be.to_host(Wim)
        for Wx in (Wre, Wim):
            be.W_from_factor(U, 2, Wx); be.W_from_factor(U, 0, Wx)
        W1 = be.to_host(Wre) + 1j * be.to_host(Wim)
        for Wx in (Wre, Wim):
            be.block_solve(D, ip_off, 0, 1, Wx); be.block_solve(D, ip_off, 1, 0, Wx)
        W2 = be.to_host(Wre) + 1j * be.to_host(Wim)
        res[name] = (M, W1, W2)
    print('q', iq, 'M', rel(res['hip'][0], res['cpu'][0]), 'herm', abs(res['cpu'][0] - res['cpu'][0].conj().T).max() / abs(res['cpu'][0]).max(),
          'W1', rel(res['hip'][1], res['cpu'][1]), 'W2', rel(res['hip'][2], res['cpu'][2]))
    # cross: hip M finished on cpu
    Mh = res['hip'][0]
    Wre = torch.from_numpy(np.ascontiguousarray(Mh.real)); Wim = torch.from_numpy(np.ascontiguousarray(Mh.imag))
    U = torch.from_numpy(out['cpu']['U'])
    for Wx in (Wre, Wim):
        ob.W_from_factor(U, 2, Wx); ob.W_from_factor(U, 0, Wx)
    W1x = Wre.numpy() + 1j * Wim.numpy()
    print('   hip M finished on cpu vs cpu', rel(W1x, res['cpu'][1]), ' vs hip', rel(W1x, res['hip'][1]))

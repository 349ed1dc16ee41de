"""Error study for an FP64-emulated P^2 G product on the low-precision matrix cores (round-2 verdict item 7; SURVEY 7.3-5 allows
it only with this study).  The Ozaki scheme splits each operand row into s slices of ~7 bits relative to the ROW maximum
(INT8 slices, exact INT32 accumulation); s slices carry 7 s bits of every entry relative to its row's largest entry.  Emulated
here in FP64 arithmetic by truncating the operands of M = w conv(Y) Y^T to b bits below the row maximum (that is what the slice
products represent; the products themselves are exact) and measuring what reaches K - for the block-Jacobi route (rows
Y' = D^-1 B: rounding amplified by cond(A')) and for the Cholesky route (rows Y = Lr^-1 B: nothing amplified).  CPU, oracle
arithmetic, diamond 2x2x2 / gth-szv on a 24^3 mesh by default.

    python tools/study_emulated_product.py [c=10] [mesh=24]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg
from pyscf_isdf_amd import gto, workloads
from pyscf_isdf_amd._common import partition_grid_by_atom
from oracle import ao as oao, isdf as oisdf, fftdf

c_isdf = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 24
basis = sys.argv[3] if len(sys.argv) > 3 else 'gth-szv'
cell = gto.diamond_supercell(2, basis, (nm, nm, nm))
nao = cell.nao_nr()
a, mesh = cell.lattice_vectors(), cell.mesh
G = int(np.prod(mesh))
coords = cell.get_uniform_grids()
rcut = gto.estimate_rcut_per_shell(cell)
ao = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, gto.get_lattice_Ls(cell, rcut=rcut.max()), rcut, rule='point')
aoT = np.ascontiguousarray(ao.T)
dm, c, occ = workloads.make_dm(cell)
owner = partition_grid_by_atom(coords, cell.atom_coords(), a)
nip = np.full(cell.natm, c_isdf * nao // cell.natm)
ref = oisdf.build_local_select_global_fit(aoT, a, mesh, owner, nip, reg_rel=1e-12)
ip = ref['ip']
P = len(ip)
own = owner[ip]
blk_off = np.append(0, np.cumsum(np.bincount(own, minlength=cell.natm)))
aoP = aoT[:, ip]
A = aoP.T.dot(aoP) ** 2
A += 1e-12 * A.diagonal().max() * np.eye(P)
B = aoP.T.dot(aoT) ** 2
w = cell.vol / G
k_exact = fftdf.get_k(ao, dm, a, mesh)
ek = lambda k: np.einsum('ij,ji', k, dm) / 4


def trunc(X, bits):
    """Keep `bits` bits of every entry below its ROW maximum (row-scaled fixed point, round to nearest)."""
    if bits is None:
        return X
    s = np.abs(X).max(axis=1, keepdims=True)
    q = s * 2.0 ** (-bits)
    return np.round(X / q) * q


def k_blockjacobi(bits):
    D = np.zeros((P, P))
    for b in range(len(blk_off) - 1):
        sl = slice(blk_off[b], blk_off[b + 1])
        D[sl, sl] = np.linalg.cholesky(A[sl, sl])
    Yp = scipy.linalg.solve_triangular(D, B, lower=True)
    V = oisdf.coulomb_V(Yp, a, mesh)
    Mp = w * trunc(V, bits).dot(trunc(Yp, bits).T)
    Mp = 0.5 * (Mp + Mp.T)
    Ap = scipy.linalg.solve_triangular(D, scipy.linalg.solve_triangular(D, A, lower=True).T, lower=True).T
    cf = scipy.linalg.cho_factor(Ap)
    Wp = scipy.linalg.cho_solve(cf, scipy.linalg.cho_solve(cf, Mp).T).T
    W = scipy.linalg.solve_triangular(D, scipy.linalg.solve_triangular(D, Wp, lower=True, trans='T').T, lower=True, trans='T').T
    return oisdf.get_k(np.ascontiguousarray(aoP.T), 0.5 * (W + W.T), dm), np.linalg.cond(Ap)


def k_cholesky(bits):
    Lr = np.linalg.cholesky(A)
    Y = scipy.linalg.solve_triangular(Lr, B, lower=True)
    V = oisdf.coulomb_V(Y, a, mesh)
    M = w * trunc(V, bits).dot(trunc(Y, bits).T)
    M = 0.5 * (M + M.T)
    Z = scipy.linalg.solve_triangular(Lr, M, lower=True, trans='T')
    W = scipy.linalg.solve_triangular(Lr, Z.T, lower=True, trans='T').T
    return oisdf.get_k(np.ascontiguousarray(aoP.T), 0.5 * (W + W.T), dm), np.linalg.cond(A)


print('diamond 2x2x2 %s, %d^3, c = %d: nao %d, P %d, E_K(exact) %.8f' % (basis, nm, c_isdf, nao, P, ek(k_exact)))
for name, fn in (('block-Jacobi', k_blockjacobi), ('Cholesky', k_cholesky)):
    k64, cond = fn(None)
    print('%s route: cond of the matrix whose inverse is applied twice = %.1e; FP64 product: dE_K(fit) %+.2e, max|dK| %.2e'
          % (name, cond, ek(k64) - ek(k_exact), abs(k64 - k_exact).max()))
    for bits in (56, 49, 42, 35, 28, 21):
        kb, _ = fn(bits)
        print('   %2d bits below the row maximum (%d INT8 slices, %2d slice products): max|K_b - K_fp64| / max|K| = %.1e,  dE_K %+.1e Eh'
              % (bits, bits // 7, (bits // 7) * (bits // 7 + 1) // 2, abs(kb - k64).max() / abs(k64).max(), ek(kb) - ek(k64)))

"""Developer check: the ISDF exchange error with PHYSICAL orbitals, for several fit variants at once.  An RHF is converged with
the plain AO-pair ISDF object (fast: the fit does not depend on the density), the reference's exact exchange is evaluated ONCE at
those orbitals on the GPU (isdf_get_k_exact), and every variant 'select:c:space[:robust]' (space = ao | occ) is rebuilt and
compared with it.  T and S by plane-wave quadrature of the AO values (tool-level plumbing, torch.fft).

    python tools/scf_orbital_scan.py diamond-444-dzvp-120 refined:12:ao refined:12:occ refined:10:occ:robust
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg
import torch
from pyscf_isdf_amd import workloads, gto
from pyscf_isdf_amd.isdf import ISDF

name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-222-dzvp-80'
variants = sys.argv[2:] or ['refined:10:ao', 'refined:10:occ']
cell = workloads.make_cell(name)
nao, nocc = cell.nao_nr(), cell.nelectron // 2
mesh = [int(x) for x in cell.mesh]
G = int(np.prod(mesh))
df = ISDF(cell, c_isdf=10, select='refined')
be = df.backend
rcut = gto.estimate_rcut_per_shell(cell)
Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
ao = be.empty((nao, G))
be.eval_ao(np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut,
           be.to_device(np.ascontiguousarray(df.grids.coords.T)), ao)
b = 2 * np.pi * np.linalg.inv(cell.lattice_vectors().T)
fr = [np.fft.fftfreq(n, 1. / n) for n in mesh]
Gv = (fr[0][:, None, None, None] * b[0] + fr[1][None, :, None, None] * b[1] + fr[2][None, None, :, None] * b[2]).reshape(-1, 3)
g2 = be.to_device(np.einsum('gi,gi->g', Gv, Gv))
F = torch.empty((nao, G), dtype=torch.complex128, device=be.device)
for r0 in range(0, nao, 64):
    r1 = min(nao, r0 + 64)
    F[r0:r1] = torch.fft.fftn(ao[r0:r1].reshape(r1 - r0, *mesh), dim=(1, 2, 3)).reshape(r1 - r0, G)
T = ((0.5 * cell.vol / G ** 2) * torch.matmul(F.conj() * g2, F.T).real).cpu().numpy()
S = ((cell.vol / G ** 2) * torch.matmul(F.conj(), F.T).real).cpu().numpy()
del F, ao, g2
torch.cuda.empty_cache()
hcore = T + df.get_pp()
torch.cuda.empty_cache()
print(name, 'nao', nao, 'nocc', nocc, 'G', G, '- T, S, hcore done', flush=True)

t0 = time.perf_counter()
df.build()
print('driver build (refined, c = 10, AO pairs) %.1f s' % (time.perf_counter() - t0), flush=True)
e, c = scipy.linalg.eigh(hcore, S)
dm = 2 * c[:, :nocc].dot(c[:, :nocc].T)
errs, focks, e_last = [], [], 0.0
for it in range(30):
    vj, vk = df.get_jk(dm)
    f = hcore + vj - 0.5 * vk
    e_el = 0.5 * np.einsum('ij,ji', hcore + f, dm)
    err = f.dot(dm).dot(S) - S.dot(dm).dot(f)
    focks.append(f); errs.append(err); focks, errs = focks[-8:], errs[-8:]
    n = len(focks)
    if n > 1:
        B = -np.ones((n + 1, n + 1)); B[n, n] = 0
        for i in range(n):
            for j in range(n):
                B[i, j] = np.vdot(errs[i], errs[j])
        rhs = np.zeros(n + 1); rhs[n] = -1
        f = sum(ci * fi for ci, fi in zip(np.linalg.lstsq(B, rhs, rcond=None)[0][:n], focks))
    print('  it %2d  E_el %.10f  dE %.2e  |err| %.1e' % (it, e_el, e_el - e_last, abs(err).max()), flush=True)
    if abs(e_el - e_last) < 1e-8 and abs(err).max() < 1e-5:
        break
    e_last = e_el
    e, c = scipy.linalg.eigh(f, S)
    dm = 2 * c[:, :nocc].dot(c[:, :nocc].T)
cocc = np.ascontiguousarray(c[:, :nocc])
occ = np.full(nocc, 2.0)


class Tagged(np.ndarray):
    pass


tdm = dm.view(Tagged)
tdm.mo_coeff, tdm.mo_occ = cocc, occ
t1 = time.perf_counter()
df.release_fit_buffers()
vk_ex = df.get_k_exact(mo_coeff=cocc, mo_occ=occ)
ek_ex = np.einsum('ij,ji', vk_ex, dm) / 4
print('exact K at the converged orbitals: %.1f s, E_K = %.10f' % (time.perf_counter() - t1, ek_ex), flush=True)
df.reset()
del df
for v in variants:
    parts = v.split(':')
    sel, cc, space = parts[0], int(parts[1]), parts[2]
    flags = parts[3:]
    robust = 'robust' in flags
    d2 = ISDF(cell, c_isdf=cc, select=sel)
    d2.pair_space = space
    d2.robust_k = robust
    if 'bj' in flags:                               # force the block-Jacobi route (no Cholesky fallback), report its probe check
        d2.fit_route = 'blockjacobi'
    if 'chol' in flags:
        d2.fit_route = 'cholesky'
    for f_ in flags:
        if f_.startswith('g') and f_[1:].isdigit():     # merge this many consecutive atom clusters per preconditioner block
            d2.bj_group = int(f_[1:])
        if f_.startswith('reg') :
            d2.reg_rel = float(f_[3:])
    try:
        t1 = time.perf_counter()
        vk = d2.get_jk(tdm, with_j=False)[1]
        dt = time.perf_counter() - t1
        if os.environ.get('SCAN_WARM'):               # second build + K with the buffers in place: what a repeated caller pays
            t1 = time.perf_counter()
            d2.build()
            vk = d2.get_jk(tdm, with_j=False)[1]
            dt = time.perf_counter() - t1
        if 'bj' in flags and d2.n_panels == 1:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                be2 = d2.backend
                aoT = be2.empty((nao, len(d2.ip)))
                be2.gather_cols(d2.ao, be2.to_device(d2.ip), aoT)
                st = d2._fit_state
                d2.bj_check = d2._bj_probe_mismatch(aoT, st['Afac'], st['Dblk'], st['ip_off'], st['theta'], G, None)
                del aoT
        print('%-30s P=%6d  build+K %6.2f s  dE_K %+.3e Eh (%.1e per atom)  max|dK| %.2e  route %s panels %d probe %s  %s'
              % (v, len(d2.ip), dt, np.einsum('ij,ji', vk, dm) / 4 - ek_ex, abs(np.einsum('ij,ji', vk, dm) / 4 - ek_ex) / cell.natm,
                 abs(vk - vk_ex).max(), d2.fit_route_used, d2.n_panels, ('%.1e' % d2.bj_check) if d2.bj_check is not None else '-', {k: round(x, 2) for k, x in d2.timings.items()}), flush=True)
    except Exception as ex:                                   # noqa: BLE001 - a variant that does not fit in memory is reported, not fatal
        print('%-26s FAILED: %s' % (v, str(ex)[:300]), flush=True)
    d2.reset()
    del d2
    torch.cuda.empty_cache()

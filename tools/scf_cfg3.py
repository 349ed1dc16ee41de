"""Developer check at the headline size (diamond 4x4x4, gth-dzvp, 120^3): converge an RHF driven by the ISDF object (robust K
unless argv[4] == 'plain'), then evaluate the reference's exact exchange ONCE at the converged orbitals (54 s) and compare the plain and the robust
ISDF exchange with it - the fit error with physical orbitals at configs[2].  Kinetic energy / overlap by plane-wave
quadrature of the AO values, on the device with torch.fft (tool-level plumbing), before the ISDF buffers are allocated."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg
import torch
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF

name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-444-dzvp-120'
cell = workloads.make_cell(name)
nao, nocc = cell.nao_nr(), cell.nelectron // 2
mesh = [int(x) for x in cell.mesh]
G = int(np.prod(mesh))

select = sys.argv[2] if len(sys.argv) > 2 else 'local'
c_isdf = int(sys.argv[3]) if len(sys.argv) > 3 else 10
drive_plain = len(sys.argv) > 4 and sys.argv[4] == 'plain'
df = ISDF(cell, c_isdf=c_isdf, select=select.split(':')[0])
if ':' in select:
    df.refine_over = float(select.split(':')[1])
df.robust_k = not drive_plain
print(name, 'select', select, 'c', c_isdf, 'SCF driven by the', 'plain' if drive_plain else 'robust', 'K', flush=True)
be = df.backend
# AO values once, for T and S
coords = df.grids.coords
from pyscf_isdf_amd import gto
rcut = gto.estimate_rcut_per_shell(cell)
Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
ao = be.empty((nao, G))
be.eval_ao(np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut, be.to_device(np.ascontiguousarray(coords.T)), ao)
b = 2 * np.pi * np.linalg.inv(cell.lattice_vectors().T)
fr = [np.fft.fftfreq(n, 1. / n) for n in mesh]
Gv = (fr[0][:, None, None, None] * b[0] + fr[1][None, :, None, None] * b[1] + fr[2][None, None, :, None] * b[2]).reshape(-1, 3)
g2 = be.to_device(np.einsum('gi,gi->g', Gv, Gv))
F = torch.empty((nao, G), dtype=torch.complex128, device=be.device)
for r0 in range(0, nao, 64):
    r1 = min(nao, r0 + 64)
    F[r0:r1] = torch.fft.fftn(ao[r0:r1].reshape(r1 - r0, *mesh), dim=(1, 2, 3)).reshape(r1 - r0, G)
T = (0.5 * cell.vol / G ** 2) * torch.matmul(F.conj() * g2, F.T).real
S = (cell.vol / G ** 2) * torch.matmul(F.conj(), F.T).real
T, S = T.cpu().numpy(), S.cpu().numpy()
del F, ao, g2
torch.cuda.empty_cache()
hcore = T + df.get_pp()                      # before the build: get_pp evaluates its own AO planes (2 x 21 GiB at this size)
torch.cuda.empty_cache()
print('T, S, hcore done', flush=True)

t0 = time.perf_counter()
df.build()
print('build (robust_k=%s) %.1f s' % (df.robust_k, 0.0), end='  ')
print('build %.1f s  %s' % (time.perf_counter() - t0, {k: round(v, 2) for k, v in df.timings.items()}), flush=True)
e, c = scipy.linalg.eigh(hcore, S)
dm = 2 * c[:, :nocc].dot(c[:, :nocc].T)
errs, focks, e_last = [], [], 0.0
for it in range(30):
    t1 = time.perf_counter()
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
        coef = np.linalg.lstsq(B, rhs, rcond=None)[0][:n]
        f = sum(ci * fi for ci, fi in zip(coef, focks))
    print('  it %2d  E_el %.10f  dE %.2e  |err| %.1e  %.2f s' % (it, e_el, e_el - e_last, abs(err).max(), time.perf_counter() - t1), flush=True)
    if abs(e_el - e_last) < 1e-8 and abs(err).max() < 1e-5:
        break
    e_last = e_el
    e, c = scipy.linalg.eigh(f, S)
    dm = 2 * c[:, :nocc].dot(c[:, :nocc].T)
cocc = c[:, :nocc]
vk_first = df.get_jk(dm, with_j=False)[1]
t1 = time.perf_counter()
df.release_fit_buffers()
vk_ex = df.get_k_exact(mo_coeff=cocc, mo_occ=np.full(nocc, 2.0))
print('exact K at the converged orbitals: %.1f s' % (time.perf_counter() - t1), flush=True)
if len(sys.argv) > 5 and sys.argv[5] == 'norobust':
    ek = lambda k: np.einsum('ij,ji', k, dm) / 4
    print('E_K exact %.10f   plain ISDF dE_K %.3e (max|dK| %.2e)   [%d atoms; robust K skipped]' %
          (ek(vk_ex), ek(vk_first) - ek(vk_ex), abs(vk_first - vk_ex).max(), cell.natm), flush=True)
    sys.exit(0)
df.robust_k = drive_plain
t1 = time.perf_counter()
df.build()
print('second build (robust_k=%s) %.1f s  %s' % (df.robust_k, time.perf_counter() - t1, {k: round(v, 2) for k, v in df.timings.items()}), flush=True)
t1 = time.perf_counter()
vk_second = df.get_jk(dm, with_j=False)[1]
print('its K: %.2f s' % (time.perf_counter() - t1), flush=True)
vk_plain, vk_rob = (vk_first, vk_second) if drive_plain else (vk_second, vk_first)
ek = lambda k: np.einsum('ij,ji', k, dm) / 4
print('E_K exact %.10f   plain ISDF dE_K %.3e (max|dK| %.2e)   robust dE_K %.3e (max|dK| %.2e)   [%d atoms]' %
      (ek(vk_ex), ek(vk_plain) - ek(vk_ex), abs(vk_plain - vk_ex).max(), ek(vk_rob) - ek(vk_ex), abs(vk_rob - vk_ex).max(), cell.natm), flush=True)

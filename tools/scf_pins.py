import sys; import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, R)
import numpy as np
import scf_helpers, test_gpu_scf as t
from pyscf_isdf_amd import multigrid as pmg
cell = t._diamond_newton_cell()
S, T = scf_helpers.overlap_kinetic_from_ft(cell)
df = pmg.MultiGridFFTDF(cell, c_isdf=6, select='global'); df.split = 'all'; df.select_tol = 0.0
hcore = T + df.get_pp(); e_nuc = scf_helpers.ewald_energy(cell)
print('levels', df.build_tasks())
def veff_lda(dm):
    n, exc, veff = pmg.nr_rks(df, 'lda,', dm, with_j=True)
    return np.asarray(veff), float(veff.ecoul), float(exc)
e, dm = scf_helpers.rks(hcore, S, veff_lda, 4, e_nuc)
print('LDA  e_tot %.12f  ref -9.7670882971475663  diff %.2e' % (e, e + 9.7670882971475663))
def veff_b88(dm):
    n, exc, veff = pmg.nr_rks(df, 'b88,', dm, with_j=True)
    return np.asarray(veff), float(veff.ecoul), float(exc)
e, dm = scf_helpers.rks(hcore, S, veff_b88, 4, e_nuc)
print('B88  e_tot %.12f  ref -9.9355341416893559  diff %.2e' % (e, e + 9.9355341416893559))
e, dm = scf_helpers.rhf(hcore, S, lambda d: df.get_jk(d, exxdiv='ewald'), 4, e_nuc)
print('RHF  e_tot %.12f  ref -10.137043711032916  diff %.2e   P=%d' % (e, e + 10.137043711032916, len(df.ip)))

# k-points [2,1,1] (test_newton.py:135-157): KRKS 'lda,' and 'b88,'
kpts = cell.make_kpts([2, 1, 1])
Sk, Tk = scf_helpers.overlap_kinetic_from_ft_kpts(cell, kpts)
dfk = pmg.MultiGridFFTDF(cell, kpts=kpts); dfk.split = 'all'
hk = Tk + np.asarray(dfk.get_pp(kpts))
for xc, ref in (('lda,', -10.307756038726733), ('b88,', -10.446717855794008)):
    def veff_k(dms, xc=xc):
        n, exc, veff = pmg.nr_rks(dfk, xc, dms, kpts=kpts, with_j=True)
        return np.asarray(veff), float(veff.ecoul), float(exc)
    e, dms = scf_helpers.krks(hk, Sk, veff_k, 4, e_nuc)
    print('KRKS %-5s e_tot %.12f  ref %.15f  diff %.2e' % (xc, e, ref, e - ref))
# the reference's stored potential constant for 'b88,' at k-points (test_multigrid.py:216-227)
import test_multigrid as tm
from oracle import pbc_tools as otools
c2, k2, dm2 = tm.reference_gga_kpts_case()
df2 = pmg.MultiGridFFTDF(c2); df2.split = 'all'
n, e, veff = pmg.nr_rks(df2, 'b88,', dm2, kpts=k2, with_j=True)
fp = otools.fp(veff)
print('fp(vxc[b88] + vj) on %d levels: %.12f%+.12fj  ref -0.05697304864467462+0.6990367789096609j  |diff| %.2e'
      % (len(df2.tasks), fp.real, fp.imag, abs(fp - (-0.05697304864467462 + 0.6990367789096609j))))

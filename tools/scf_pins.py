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

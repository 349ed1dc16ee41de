import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, R)
import numpy as np, torch
import test_multigrid as t
from pyscf_isdf_amd import multigrid as pmg
from oracle import multigrid as omg
cell = t.cell_he_split()
a, mesh = cell.lattice_vectors(), cell.mesh
df = pmg.MultiGridFFTDF(cell); df.split = 'all'
dm = t.make_dm(cell)
tasks = t.as_tasks(df.build_tasks())
be = df.backend
for wj, rj in ((False, False), (True, False), (False, True)):
    n, e, veff = pmg.nr_rks(df, 'b88,', dm, with_j=wj, return_j=rj)
    n0, e0, v0, ec0 = omg.nr_rks_b88(tasks, cell._atm, dm, a, mesh, with_j=wj)
    print('with_j', wj, 'return_j', rj, 'dveff', abs(veff - v0).max())
# gemm accumulate check on level shapes
for lv in df.tasks:
    ldp = -(-lv.ngrids // 32) * 32
    rng = np.random.default_rng(1)
    A = rng.standard_normal((lv.nT, ldp)); A[:, lv.ngrids:] = 0
    B = rng.standard_normal((lv.nT, ldp)); B[:, lv.ngrids:] = 0
    s = rng.standard_normal(ldp); s[lv.ngrids:] = 0
    dA, dB, ds = be.to_device(A), be.to_device(B), be.to_device(s)
    V = be.empty((lv.nH, lv.nT))
    be.gemm_nt(dA[:lv.nH], dA, V, kscale=ds)
    be.gemm_nt(dA[:lv.nH], dB, V, beta=1.0, kscale=ds)
    be.gemm_nt(dB[:lv.nH], dA, V, beta=1.0, kscale=ds)
    ref = A[:lv.nH].dot((A * s).T) + A[:lv.nH].dot((B * s).T) + B[:lv.nH].dot((A * s).T)
    print(lv, 'gemm accumulate err', abs(be.to_host(V) - ref).max())

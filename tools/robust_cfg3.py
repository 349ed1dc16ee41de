"""Developer timing / accuracy: build + get_jk with robust_k at a workload (default configs[2]) for several values of the
fit's diagonal shift (the robust form is variational: a regularisation bias enters quadratically, rounding noise of the
ill-conditioned fit enters linearly - so a larger shift than the plain ISDF default pays)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF
name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-444-dzvp-120'
regs = [float(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1e-12]
cell = workloads.make_cell(name)
dm, c, occ = workloads.make_dm(cell)
df = ISDF(cell, c_isdf=10)
df.robust_k = True
ref = None
for reg in regs:
    df.reg_rel = reg
    t0 = time.perf_counter(); df.build(); df.backend.synchronize(); t1 = time.perf_counter()
    vj, vk = df.get_jk(dm); df.backend.synchronize(); t2 = time.perf_counter()
    if ref is None:
        ref = df.get_k_exact(mo_coeff=c, mo_occ=occ)
    ek, ek0 = np.einsum('ij,ji', vk, dm) / 4, np.einsum('ij,ji', ref, dm) / 4
    print('reg_rel %.0e (used %.0e): build %.2f s  get_jk %.2f s  EK(robust) %.10f  EK(exact) %.10f  dEK %.2e  max|dK| %.2e' %
          (reg, df.reg_used, t1 - t0, t2 - t1, ek, ek0, ek - ek0, abs(vk - ref).max()), flush=True)

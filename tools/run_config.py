"""Developer helper: build + get_jk on one workload, print per-stage wall times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF

name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-222-dzvp-80'
select = sys.argv[2] if len(sys.argv) > 2 else 'local'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cell = workloads.make_cell(name)
dm, c, occ = workloads.make_dm(cell)
print(name, 'natm', cell.natm, 'nao', cell.nao_nr(), 'mesh', cell.mesh, 'select', select, flush=True)
df = ISDF(cell, c_isdf=10, select=select)
for it in range(reps):
    t0 = time.perf_counter()
    df.build()
    vj, vk = df.get_jk(dm)
    df.backend.synchronize()
    t1 = time.perf_counter()
    print('iter %d total %.3f s  P=%d  reg=%g' % (it, t1 - t0, len(df.ip), df.reg_used))
    for k, v in df.timings.items():
        print('   %-18s %8.3f s' % (k, v))
    print('   EJ %.10f  EK %.10f' % (np.einsum('ij,ji', vj, dm) / 2, np.einsum('ij,ji', vk, dm) / 4), flush=True)

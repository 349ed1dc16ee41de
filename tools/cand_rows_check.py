"""Does skipping the AO rows that vanish on a block change the selection?  Same points expected; candidate-stage time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF
name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-444-dzvp-120'
cc = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cell = workloads.make_cell(name)
dm = workloads.make_dm(cell)[0]
res = {}
df = ISDF(cell, c_isdf=cc, select='refined')
for flag in (False, True, False, True):
    df.cand_skip_zero_rows = flag
    t0 = time.perf_counter()
    df.build()
    vk = df.get_jk(dm, with_j=False)[1]
    df.backend.synchronize()
    dt = time.perf_counter() - t0
    print('skip_zero_rows=%s: build+K %.2f s  candidates %.3f s  rows kept %s  E_K %.10f' % (flag, dt, df.timings['S2_select_candidates'], getattr(df, '_cand_rows_kept', None), np.einsum('ij,ji', vk, dm) / 4), flush=True)
    if flag in res:
        print('   same points as the first run with this flag:', np.array_equal(res[flag][0], df.ip))
    res.setdefault(flag, (df.ip.copy(), vk.copy(), df))
print('points identical with and without:', np.array_equal(res[False][0], res[True][0]), ' max|dK| between them: %.2e' % abs(res[False][1] - res[True][1]).max())

"""Where the cold first build spends its extra seconds: stage wall times of the first and second build in a fresh process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t00 = time.perf_counter()
import numpy as np
import torch
from pyscf_isdf_amd import ISDF, gto
print('imports %.2f s' % (time.perf_counter() - t00), flush=True)
cell = gto.diamond_supercell(4, mesh=(120,) * 3)
nao = cell.nao_nr()
dm = np.eye(nao)
for it in range(2):
    t0 = time.perf_counter()
    df = ISDF(cell, c_isdf=12) if it == 0 else df
    if it == 0:
        tb = time.perf_counter()
        be = df.backend
        print('backend/handle creation %.2f s' % (time.perf_counter() - tb), flush=True)
    df._built = False
    df.build()
    df.get_jk(dm)
    torch.cuda.synchronize()
    print('build %d: %.2f s  %s' % (it, time.perf_counter() - t0, {k: round(v, 2) for k, v in df.timings.items()}), flush=True)

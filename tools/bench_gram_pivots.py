"""Pivot step of the Gram selection (isdf_select_ip_gram) at the headline's size for the workgroup widths of option
"gram_pivot_tpb": same pivots expected, time per pivot from the library's event pairs.  GPU only.

    python tools/bench_gram_pivots.py [m=39936] [N=1664]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_isdf_amd.backend import HipBackend

m = int(sys.argv[1]) if len(sys.argv) > 1 else 39936
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1664
be = HipBackend()
g = torch.Generator(device=be.device).manual_seed(1)
X = torch.randn((N, m), dtype=torch.float64, device=be.device, generator=g)
X *= torch.exp(-3.0 * torch.rand((1, m), dtype=torch.float64, device=be.device, generator=g))      # spread of the diagonal
S = X.T @ X
ref = None
be.prof_enable(True)
panels = [int(x) for x in os.environ.get('PANELS', '256').split(',')]
for tpb, panel in [(t, q) for q in panels for t in (256, 128, 64)]:
    be.set_option('gram_pivot_tpb', tpb)
    A = S * S
    piv = be.empty((m // 2,), dtype=torch.int64)
    be.synchronize()
    be.prof_reset()
    t0 = time.perf_counter()
    rank = be.select_ip_gram(A, m // 2, -1.0, 1e-10, piv, panel=panel)
    be.synchronize()
    dt = time.perf_counter() - t0
    pr = be.prof_results()
    st = pr.get('gram_pivot_step_kernel[byte]', dict(ms=0.0, launches=0))
    ph = be.to_host(piv)
    same = None if ref is None else bool(np.array_equal(ref, ph))
    if ref is None:
        ref = ph
    print('panel %3d ' % panel + 'tpb %3d: rank %d  whole call %.3f s  pivot steps %.1f ms (%.2f us per pivot)  same pivots as the first run: %s'
          % (tpb, rank, dt, st['ms'], 1e3 * st['ms'] / max(1, rank), same), flush=True)
    del A

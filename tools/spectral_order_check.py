"""Spectral W with the packed points in shells of |G| (w_sort_bins): probe mismatch, E_K, time of the transform + packing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF
name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-222-dzvp-80'
cc = int(sys.argv[2]) if len(sys.argv) > 2 else 12
binsl = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '0,16,256,4096').split(',')]
cell = workloads.make_cell(name)
dm = workloads.make_dm(cell)[0]
ref = None
df = ISDF(cell, c_isdf=cc, select='refined')
df.w_spectral_max_c, df.bj_max_c, df.w_spectral_check_tol = 99, 99, 1.0
for bins in [-1] + binsl + binsl:
    df.w_spectral = bins >= 0
    df.w_sort_bins = max(bins, 0)
    df._built = False
    df.backend.prof_enable(True); df.backend.prof_reset()
    t0 = time.perf_counter()
    vk = df.get_jk(dm, with_j=False)[1]
    df.backend.synchronize()
    dt = time.perf_counter() - t0
    pr = df.backend.prof_results()
    sr = pr.get('spectral_rows_own[byte]', dict(ms=0.0))
    ek = np.einsum('ij,ji', vk, dm) / 4
    if ref is None:
        ref = (vk.copy(), ek)
    print('%-22s E_K %.12f (%+.3e vs classic)  max|dK| %.2e  probe %.2e  fraction %s  panels %d  build+K %.2f s  transform+pack %.0f ms' % (
        'classic' if bins < 0 else 'spectral, %d shells' % bins, ek, ek - ref[1], abs(vk - ref[0]).max(), df.bj_check or 0, df.w_spectral_fraction,
        df.n_panels, dt, sr['ms']), flush=True)

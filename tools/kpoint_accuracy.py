"""k-point ISDF exchange against the reference's exact k-point exchange on the GPU (isdf_get_k_exact_kpt), for a scan over
k_ip_factor / c.  Full matrices where that takes minutes at most; a sample of AO rows (argv[4] = number of rows) at sizes where
the exact path would take an hour (MgO 3x3x3: 64 x 729 x 216 complex FFT pairs of 96^3).

    python tools/kpoint_accuracy.py mgo-222-dzvp-k222 10 2,3,4 [nrows]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF

name = sys.argv[1] if len(sys.argv) > 1 else 'mgo-222-dzvp-k222'
c_isdf = int(sys.argv[2]) if len(sys.argv) > 2 else 10
facs = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [2, 4]
nrows = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cell = workloads.make_cell(name)
if os.environ.get('MESH'):                     # e.g. MESH=75: an odd mesh (no Nyquist planes)
    from pyscf_isdf_amd import gto as _gto
    n_ = int(os.environ['MESH'])
    cell = workloads.mgo_supercell(2 if '222' in name else 3, 'gth-dzvp', (n_, n_, n_))
kpts = workloads.make_kpts(name, cell)
nk, nao = len(kpts), cell.nao_nr()
rng = np.random.default_rng(20240203)
occ = np.zeros(nao); occ[:cell.nelectron // 2] = 2
cs, dms = [], []
for k in range(nk):
    c = np.linalg.qr(rng.standard_normal((nao, nao)) + 1j * rng.standard_normal((nao, nao)))[0]
    cs.append(c); dms.append((c * occ).dot(c.conj().T))
dms, cs, occs = np.array(dms), np.array(cs), np.tile(occ, (nk, 1))
print(name, 'nao', nao, 'nk', nk, 'nocc', int((occ > 0).sum()), 'G', int(np.prod(cell.mesh)), flush=True)
df = ISDF(cell, kpts=kpts, c_isdf=c_isdf, select='refined')
t0 = time.perf_counter()
rows = None
if nrows:
    i0 = nao // 2 - nrows // 2
    rows = (i0, nrows)
vk_ex = df.get_k_exact(dms, mo_coeff=cs, mo_occ=occs, rows=rows)
print('exact k-point exchange%s: %.1f s' % ('' if rows is None else ' (rows %d..%d of every k-point)' % (rows[0], rows[0] + rows[1]), time.perf_counter() - t0), flush=True)
df.reset()
for fac in facs:
    df = ISDF(cell, kpts=kpts, c_isdf=c_isdf, select='refined')
    df.k_ip_factor = fac
    if os.environ.get('PAIR'):                      # PAIR=1: force the +-q pairing (round-2 behaviour); default 'auto'
        df.kpt_pair_q = True
    df.robust_k = bool(os.environ.get('ROBUST'))          # Dunlap's correction at k-points (V^q recomputed per K)
    t0 = time.perf_counter()
    vk = df.get_jk(dms, kpts=kpts, with_j=False)[1]
    dt = time.perf_counter() - t0
    if rows is None:
        de = np.einsum('kij,kji', vk - vk_ex, dms).real / 4 / nk
        print('%sc=%d k_ip_factor=%d P=%d  build+K %.1f s  dE_K %+.3e Eh per cell  max|dK| %.2e  (E_K exact %.8f)'
              % ('robust K, ' if df.robust_k else '', c_isdf, fac, len(df.ip), dt, de, abs(vk - vk_ex).max(), np.einsum('kij,kji', vk_ex, dms).real / 4 / nk), flush=True)
    else:
        sub = vk[:, rows[0]:rows[0] + rows[1]]
        print('c=%d k_ip_factor=%d P=%d  build+K %.1f s  max|dK| on the sampled rows %.2e (max|K| there %.3f)'
              % (c_isdf, fac, len(df.ip), dt, abs(sub - vk_ex).max(), abs(vk_ex).max()), flush=True)
    df.reset()

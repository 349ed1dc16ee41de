"""Developer helper: k-point ISDF build + get_jk on one workload, per-stage wall times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF

name = sys.argv[1] if len(sys.argv) > 1 else 'mgo-222-dzvp-k222'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cell = workloads.make_cell(name)
kpts = workloads.make_kpts(name, cell)
nk, nao = len(kpts), cell.nao_nr()
rng = np.random.default_rng(20240203)
dms = []
for k in range(nk):
    c = np.linalg.qr(rng.standard_normal((nao, nao)) + 1j * rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:cell.nelectron // 2] = 2
    dms.append((c * occ).dot(c.conj().T))
dms = np.array(dms)
print(name, 'natm', cell.natm, 'nao', nao, 'mesh', cell.mesh, 'nk', nk, flush=True)
df = ISDF(cell, kpts=kpts, c_isdf=10, select='local')
for it in range(reps):
    t0 = time.perf_counter()
    df.build()
    vj, vk = df.get_jk(dms, kpts=kpts)
    df.backend.synchronize()
    print('iter %d total %.3f s  P=%d  nq=%d (built %d)' % (it, time.perf_counter() - t0, len(df.ip), len(df._qs), len(df._Wq)))
    for k, v in df.timings.items():
        print('   %-18s %8.3f s' % (k, v))
    print('   EJ %.10f  EK %.10f  herm(K) %.2e' % (np.einsum('kij,kji', vj, dms).real / 2 / nk, np.einsum('kij,kji', vk, dms).real / 4 / nk,
                                                  abs(vk - vk.conj().transpose(0, 2, 1)).max()), flush=True)

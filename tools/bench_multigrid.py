"""Multigrid J against the plain FFTDF-formula J on the device, configs[2] shape by default (diamond 4x4x4, gth-dzvp, 120^3).
Usage: python tools/bench_multigrid.py [n_supercell] [mesh] [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_isdf_amd import gto, multigrid as pmg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = int(sys.argv[2]) if len(sys.argv) > 2 else 120
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cell = gto.diamond_supercell(n, mesh=(m,) * 3)
nao = cell.nao_nr()
rng = np.random.default_rng(0)
dm = rng.standard_normal((nao, nao)) * 0.01
dm = dm + dm.T + np.eye(nao)
df = pmg.MultiGridFFTDF(cell)
be = df.backend
print('ladder:', df.build_tasks(), flush=True)


def timed(fn, label):
    fn(); be.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    be.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%-46s %.3f s' % (label, dt), flush=True)
    return out, dt


# plain J: collocation on the dense mesh + isdf_get_j, what ISDF.get_jk(with_k=False) runs after its build
rcut = gto.estimate_rcut_per_shell(cell)
Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
G = m ** 3
coords_soa = be.to_device(np.ascontiguousarray(cell.get_uniform_grids().T))
ao = be.empty((nao, G))
mesh = np.asarray(cell.mesh, dtype=np.int32)
a = cell.lattice_vectors()


def plain(with_ao):
    if with_ao:
        be.eval_ao(cell._atm, cell._bas, cell._env, Ls, rcut, coords_soa, ao)
    d_vj = be.empty((1, nao, nao))
    be.get_j(ao, G, mesh, a, be.to_device(dm[None]), d_vj)
    return be.to_host(d_vj)[0]


ref, t_plain_ao = timed(lambda: plain(True), 'plain J incl. collocation')
_, t_plain = timed(lambda: plain(False), 'plain J, AOs resident (ISDF.get_jk)')
del ao
torch.cuda.empty_cache()
for frac, label in ((0.0, 'multigrid J, levels collocated in both passes'), (0.25, 'multigrid J, level AOs resident')):
    df.ao_cache_fraction = frac
    df._level_cache = {}
    vj, t_mg = timed(lambda: df.get_jk(dm, with_k=False)[0], label)
    be.prof_enable(True); be.prof_reset()
    df.get_jk(dm, with_k=False); be.synchronize()
    for k, v in sorted(be.prof_results().items(), key=lambda kv: -kv[1]['ms']):
        print('      %-40s launches %4d  %8.2f ms' % (k, v['launches'], v['ms']))
    be.prof_enable(False)
    print('   max |J_multigrid - J_plain| = %.3e   (|J|max %.3f)' % (abs(vj - ref).max(), abs(ref).max()), flush=True)

# J + XC in one pass pair (what an SCF iteration of a (hybrid) functional asks of this object; K = 0.05 s from the ISDF fit)
df.ao_cache_fraction = 0.25
df._level_cache = {}
torch.cuda.empty_cache()
for xc in ('lda,', 'b88,'):
    out, t = timed(lambda: pmg.nr_rks(df, xc, dm, with_j=True), "nr_rks('%s', with_j=True): J + XC potential matrix" % xc)
    print('   nelec %.8f  exc %.8f  ecoul %.8f' % (out[0], out[1], out[2].ecoul), flush=True)
    df._level_cache = {}
    torch.cuda.empty_cache()

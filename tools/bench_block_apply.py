"""Micro-benchmark of isdf_block_apply (Y' = D^-1 B over grid columns): the register-resident kernel against the round-2 kernel,
at the headline block structure (128 blocks of 156 rows) on a slice of grid columns."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_isdf_amd.backend import HipBackend

be = HipBackend(0)
nblk, mb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 156)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 400000
P = nblk * mb
off = (np.arange(nblk + 1) * mb).astype(np.int32)
rng = np.random.default_rng(0)
D = np.zeros((P, P))
for b in range(nblk):
    M = rng.standard_normal((mb, mb))
    D[off[b]:off[b + 1], off[b]:off[b + 1]] = np.linalg.cholesky(M.dot(M.T) + mb * np.eye(mb))
dD = be.to_device(D)
dI = be.empty((P, P))
be.block_invert(dD, off, dI)
X0 = torch.randn((P, n), dtype=torch.float64, device=be.device)
ref = None
waves = [int(x) for x in os.environ.get('WAVES', '4').split(',')]
for opt in [0] + [1] * len(waves):
    be.set_option('block_apply_reg', opt)
    if opt:
        wv = waves.pop(0)
        be.set_option('block_apply_waves', wv)
    X = None
    X = X0.clone()
    be.block_apply(dI, off, X)
    be.synchronize()
    if ref is None:
        ref = X.clone()
    else:
        print('   max|difference| to the round-2 kernel: %.2e' % float((X - ref).abs().max()), flush=True)
    ts = []
    for _ in range(5):
        be.synchronize(); t0 = time.perf_counter()
        be.block_apply(dI, off, X)
        be.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print('block_apply_reg=%d%s: %d blocks x %d rows x %d columns: %.2f ms  %.2f TB/s algorithmic (16 P n bytes)' % (opt, (' waves=%d' % wv) if opt else '', nblk, mb, n, t * 1e3, 16.0 * P * n / t / 1e12), flush=True)

"""Micro-benchmark: hand-written FP64 MFMA NN GEMM (pair-density rows, C = A B with K = the AO count) vs rocBLAS (torch.matmul)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscf_isdf_amd.backend import HipBackend
be = HipBackend(0)
# (rows of the panel, grid points, AOs): configs[2] diamond 4x4x4 gth-dzvp full panel / 512-row recompute batch; water-64-like
shapes = [(4096, 1728000, 1664), (512, 1728000, 1664), (9984, 1728000, 1664), (2048, 1259712, 1472)]
for M, N, K in shapes:
    A = torch.randn(M, K, dtype=torch.float64, device=be.device)
    B = torch.randn(K, N, dtype=torch.float64, device=be.device)
    C = torch.empty(M, N, dtype=torch.float64, device=be.device)
    for name, fn in (('mfma_nn', lambda: be.gemm_nn(A, B, C)), ('rocblas', lambda: torch.matmul(A, B, out=C))):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('%-8s M=%d N=%d K=%d: %.2f ms  %.1f TF/s' % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    be.gemm_nn(A, B, C)
    sub = slice(0, 200000)
    ref = torch.matmul(A, B[:, sub])
    print('   max rel diff vs rocblas: %.2e' % ((C[:, sub] - ref).abs().max() / ref.abs().max()).item())
    del A, B, C, ref

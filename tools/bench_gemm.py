"""Micro-benchmark: hand-written FP64 MFMA NT GEMM vs rocBLAS (through torch.matmul) on the W / vj shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscf_isdf_amd.backend import HipBackend
be = HipBackend(0)
shapes = [(384, 16640, 1728000 // 4), (512, 2080, 512000), (1024, 2080, 512000), (1664, 1664, 1728000 // 2), (208, 208, 512000),
          (384, 7020, 884736)]
for M, N, K in shapes:
    A = torch.randn(M, K, dtype=torch.float64, device=be.device)
    B = torch.randn(N, K, dtype=torch.float64, device=be.device)
    C = torch.empty(M, N, dtype=torch.float64, device=be.device)
    for name, fn in (('mfma_nt', lambda: be.gemm_nt(A, B, C)), ('rocblas', lambda: torch.matmul(A, B.T, out=C))):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('%-8s M=%d N=%d K=%d: %.2f ms  %.1f TF/s' % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    ref = torch.matmul(A, B.T)
    be.gemm_nt(A, B, C)
    print('   max rel diff vs rocblas: %.2e' % ((C - ref).abs().max() / ref.abs().max()).item())
    del A, B, C, ref

"""Micro-benchmark of the FP64 MFMA NT GEMM variants on the W shape (run once per ISDF_GEMM_VARIANT: the choice is read once)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscf_isdf_amd.backend import HipBackend
be = HipBackend(0)
for M, N, K in [(512, 16640, 1728000 // 2), (512, 8320, 1728000), (512, 2080, 512000), (1664, 1664, 1728000)]:
    A = torch.randn(M, K, dtype=torch.float64, device=be.device)
    B = torch.randn(N, K, dtype=torch.float64, device=be.device)
    C = torch.empty(M, N, dtype=torch.float64, device=be.device)
    be.gemm_nt(A, B, C); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps): be.gemm_nt(A, B, C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ref = torch.matmul(A[:, :K // 8], B[:, :K // 8].T)
    be.gemm_nt(A[:, :K // 8], B[:, :K // 8], C)
    print('variant %s  M=%d N=%d K=%d: %.2f ms  %.1f TF/s   max rel diff vs rocblas (K/8): %.1e' %
          (os.environ.get('ISDF_GEMM_VARIANT', 'auto'), M, N, K, ms, 2.0 * M * N * K / ms / 1e9,
           ((C - ref).abs().max() / ref.abs().max()).item()), flush=True)
    del A, B, C, ref

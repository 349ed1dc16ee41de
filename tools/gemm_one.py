"""One shape of the MFMA NT GEMM, a few launches (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscf_isdf_amd.backend import HipBackend
be = HipBackend(0)
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (384, 16640, 432000)
A = torch.randn(M, K, dtype=torch.float64, device=be.device)
B = torch.randn(N, K, dtype=torch.float64, device=be.device)
C = torch.empty(M, N, dtype=torch.float64, device=be.device)
for _ in range(3):
    be.gemm_nt(A, B, C)
torch.cuda.synchronize()
print('done')

"""Developer experiment: shader clock and power while the FP64 MFMA NT GEMM (and, for comparison, rocBLAS dgemm on its
best shape) keeps the chip busy - is the 0.82-of-peak plateau a clock/power effect?"""
import sys, os, time, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscf_isdf_amd.backend import HipBackend
be = HipBackend(0)
samples = []
stop = False


def poll():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.perf_counter(), out.strip().splitlines()[-1] if out.strip() else ''))
        except Exception as e:                     # noqa: BLE001
            samples.append((time.perf_counter(), 'ERR %s' % e))
        time.sleep(0.3)


def run(name, fn, reps):
    global samples
    samples = []
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(name, '%.2f s' % (t1 - t0), flush=True)
    for t, s in samples:
        if t0 + 0.5 < t < t1:
            print('   ', '%.1f' % (t - t0), s[:200])


th = threading.Thread(target=poll, daemon=True)
th.start()
time.sleep(1.0)
print('idle:', samples[-1][1][:200] if samples else None)
hdr = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True).stdout.strip().splitlines()
print('header:', hdr[0][:300] if hdr else None)
M, N, K = 512, 8320, 1728000
A = torch.randn(M, K, dtype=torch.float64, device=be.device)
B = torch.randn(N, K, dtype=torch.float64, device=be.device)
C = torch.empty(M, N, dtype=torch.float64, device=be.device)
run('mfma_nt 512x8320x1728000 x12 (%.1f TFLOP each)' % (2.0 * M * N * K / 1e12), lambda: be.gemm_nt(A, B, C), 12)
del A, B, C
P, nao, G = 16640, 1664, 432000
a = torch.randn(P, nao, dtype=torch.float64, device=be.device)
b = torch.randn(nao, G, dtype=torch.float64, device=be.device)
c = torch.empty(P, G, dtype=torch.float64, device=be.device)
run('rocblas dgemm 16640x432000x1664 x8 (%.1f TFLOP each)' % (2.0 * P * nao * G / 1e12), lambda: torch.matmul(a, b, out=c), 8)
stop = True

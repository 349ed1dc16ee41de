"""Probe: what INT8 GEMM rate does the library stack reach on this GPU for the shape of the W product (NT, K long)?  Sets the
expectation for an FP64-emulated (Ozaki) product: FP64-equivalent rate = INT8 rate / number of slice products."""
import time, torch
dev = torch.device('cuda', 0)
for M, N, K in ((512, 19968, 131072), (2048, 19968, 131072), (4096, 4096, 131072), (8192, 8192, 8192)):
    a = torch.randint(-127, 127, (M, K), dtype=torch.int8, device=dev)
    b = torch.randint(-127, 127, (N, K), dtype=torch.int8, device=dev)
    try:
        bt = b.t()
        c = torch._int_mm(a, bt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            c = torch._int_mm(a, bt)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 5
        print('int8 NT %5d x %5d x %6d: %.2f ms  %.2f POP/s' % (M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e15), flush=True)
    except Exception as e:
        print('int8 %d x %d x %d failed: %s' % (M, N, K, str(e)[:200]), flush=True)
    del a, b
# bf16 for comparison
for M, N, K in ((2048, 19968, 131072),):
    a = torch.randn((M, K), dtype=torch.bfloat16, device=dev)
    b = torch.randn((N, K), dtype=torch.bfloat16, device=dev)
    c = a @ b.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        c = a @ b.t()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5
    print('bf16 NT %5d x %5d x %6d: %.2f ms  %.2f PFLOP/s' % (M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e15), flush=True)

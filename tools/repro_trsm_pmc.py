"""Repro for the note in pyscf_isdf_amd/csrc/trsm.hip / include/mi355_isdf.h: rocBLAS dtrsm with a 1.7M-column right-hand side
(the Cholesky fit route at configs[2]) under `rocprofv3 --pmc`.  Prints the call's exact arguments and rocBLAS's own
workspace requirement (device-memory size query) first, then runs the solve once through the library and checks it.
Run plain, and ONCE under the profiler with the interpreter directly after `--`:
    python3 tools/repro_trsm_pmc.py
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/trsm_pmc -- python3 tools/repro_trsm_pmc.py
"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_isdf_amd.backend import HipBackend

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1728000
be = HipBackend(0)
print('rb_left(m=%d, n=%d): rocblas_dtrsm(side=right, uplo=upper, transA=none, diag=non_unit, M=%d, N=%d, lda=%d, ldb=%d); '
      'M*N = %d elements (int32 max 2147483647), B = %.1f GiB' % (m, n, n, m, m, n, m * n, 8.0 * m * n / 2 ** 30), flush=True)
# rocBLAS's own statement of the workspace this call needs
rb = ctypes.CDLL('librocblas.so', mode=ctypes.RTLD_GLOBAL)
hdl = ctypes.c_void_p()
assert rb.rocblas_create_handle(ctypes.byref(hdl)) == 0
rb.rocblas_start_device_memory_size_query.argtypes = [ctypes.c_void_p]
rb.rocblas_stop_device_memory_size_query.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
rb.rocblas_dtrsm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
rng = np.random.default_rng(0)
Lh = np.tril(rng.standard_normal((m, m))) * 0.01 + np.eye(m)
L = be.to_device(Lh)
X = be.empty((m, n))
X.normal_()
X0 = X[:, :4096].clone()
one = ctypes.c_double(1.0)
st = rb.rocblas_start_device_memory_size_query(hdl)
# rocblas enums: side_right = 142, fill_upper = 121, operation_none = 111, diagonal_non_unit = 131
s2 = rb.rocblas_dtrsm(hdl, 142, 121, 111, 131, n, m, ctypes.byref(one), ctypes.c_void_p(L.data_ptr()), m,
                      ctypes.c_void_p(X.data_ptr()), n)
size = ctypes.c_size_t(0)
s3 = rb.rocblas_stop_device_memory_size_query(hdl, ctypes.byref(size))
print('rocBLAS device-memory size query: start %d, dtrsm %d, stop %d -> workspace %.3f GiB' % (st, s2, s3, size.value / 2 ** 30), flush=True)
rb.rocblas_destroy_handle(hdl)
free, total = torch.cuda.mem_get_info()
print('device memory free %.1f GiB of %.1f GiB before the solve' % (free / 2 ** 30, total / 2 ** 30), flush=True)
# the library call the fit makes: X <- L^-1 X on all n columns, then back with L^-T ... (isdf_factor_solve = both solves)
be.factor_solve(L, X)
be.synchronize()
print('isdf_factor_solve (rocBLAS dtrsm x2) returned', flush=True)
ref = np.linalg.solve(Lh.dot(Lh.T), be.to_host(X0))
err = abs(be.to_host(X[:, :4096]) - ref).max() / abs(ref).max()
print('max relative error on the first 4096 columns: %.2e' % err, flush=True)
assert err < 1e-9
print('REPRO DONE', flush=True)

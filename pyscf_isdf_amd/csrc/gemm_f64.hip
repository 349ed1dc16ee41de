// Dense FP64 products for the ISDF path.
//  * gemm_rm: row-major wrapper over rocBLAS dgemm for the well-shaped products (rocBLAS reaches
//    70-74 TF/s of the 78.6 TF/s FP64 MFMA peak on them: profiles/r01_probe_rocblas_hipfft_mfma64.log).
//  * gemm_nt_f64: C = alpha * A * B^T + beta * C with K contiguous in BOTH operands and K >> M, N —
//    the shape of W = V Theta^T (K = ngrids) and of vj = ao (v.ao)^T.  rocBLAS runs this shape at
//    1-13 TF/s (same log), so it gets a hand-written v_mfma_f64_16x16x4_f64 split-K kernel here.
#include "common.h"

int gemm_rm(isdf_handle h, char opA, char opB, int64_t M, int64_t N, int64_t K, double alpha,
            const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
            int64_t ldc) {
  // Row-major C = op(A) op(B)  <=>  column-major C^T = op(B)^T op(A)^T.
  ARG_CHECK(h, M < 2147483647LL && N < 2147483647LL && K < 2147483647LL && lda < 2147483647LL &&
                   ldb < 2147483647LL && ldc < 2147483647LL);
  const rocblas_operation ta = (opA == 'N') ? rocblas_operation_none : rocblas_operation_transpose;
  const rocblas_operation tb = (opB == 'N') ? rocblas_operation_none : rocblas_operation_transpose;
  BLAS_TRY(h, rocblas_dgemm(h->blas, tb, ta, (rocblas_int)N, (rocblas_int)M, (rocblas_int)K, &alpha, B,
                            (rocblas_int)ldb, A, (rocblas_int)lda, &beta, C, (rocblas_int)ldc));
  return ISDF_OK;
}

int gemm_nt_f64(isdf_handle h, int M, int N, int64_t K, double alpha, const double* A, int64_t lda,
                const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  return gemm_rm(h, 'N', 'T', M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
}

// S3: least-squares fit of the interpolation vectors Theta.
//   (a) from the selection's own Cholesky rows:  Theta = T^-1 L,  T = L[:, piv]  (upper triangular)
//   (b) for an arbitrary point set:  Theta = [(aoP aoP^T)^2]^-1 (aoP ao)^2  by Cholesky
// The triangular solves run over all G right-hand sides in place on the (P, G) row-major array.
#include "common.h"

namespace {

// T[t*k + s] = (s >= t) ? L[t*ldL + piv[s]] : 0
__global__ void gather_T_kernel(const double* __restrict__ L, int64_t ldL,
                                const int64_t* __restrict__ piv, int k, double* __restrict__ T) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (s >= k) return;
  T[(int64_t)t * k + s] = (s >= t) ? L[(int64_t)t * ldL + piv[s]] : 0.0;
}

__global__ void square_kernel(double* __restrict__ x, int64_t rows, int64_t cols, int64_t ld) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c >= cols) return;
  const double v = x[r * ld + c];
  x[r * ld + c] = v * v;
}

__global__ void transpose_gather_kernel(const double* __restrict__ ao, int64_t ld,
                                        const int64_t* __restrict__ ip, int P, int nao,
                                        double* __restrict__ aoP) {
  // aoP[p*nao + mu] = ao[mu*ld + ip[p]]
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int mu = blockIdx.y;
  if (p >= P) return;
  aoP[(int64_t)p * nao + mu] = ao[(int64_t)mu * ld + ip[p]];
}

}  // namespace

extern "C" int isdf_fit_from_chol(isdf_handle h, double* d_L, int k, int64_t m, int64_t ldL,
                                  const int64_t* d_piv) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_L && d_piv && k > 0 && m >= k && ldL >= m);
  double* T = (double*)isdf_ws(h, "fit_T", sizeof(double) * (size_t)k * k);
  if (!T) return ISDF_ERR_HIP;
  dim3 grid((unsigned)cdiv(k, 256), (unsigned)k);
  ARG_CHECK(h, k <= 65535);
  hipLaunchKernelGGL(gather_T_kernel, grid, dim3(256), 0, h->stream, d_L, ldL, d_piv, k, T);
  KERNEL_CHECK(h);
  // Row-major Theta = T^-1 L  <=>  column-major X * (T as column-major = T^T, lower) = L^T.
  const double one = 1.0;
  ARG_CHECK(h, m < (int64_t)2147483647 && ldL < (int64_t)2147483647);
  BLAS_TRY(h, rocblas_dtrsm(h->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none,
                            rocblas_diagonal_non_unit, (rocblas_int)m, (rocblas_int)k, &one, T,
                            (rocblas_int)k, d_L, (rocblas_int)ldL));
  return ISDF_OK;
}

extern "C" int isdf_fit_global(isdf_handle h, const double* d_ao, int nao, int64_t ngrids, int64_t ld,
                               const int64_t* d_ip, int P, double* d_theta, int64_t ldt,
                               double* d_aoP) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_ao && d_ip && d_theta && d_aoP && nao > 0 && P > 0 && ngrids >= P && ld >= ngrids && ldt >= ngrids);
  ARG_CHECK(h, nao <= 65535 && P <= 65535 && ldt < (int64_t)2147483647);
  // aoP (P, nao)
  hipLaunchKernelGGL(transpose_gather_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)nao), dim3(256), 0,
                     h->stream, d_ao, ld, d_ip, P, nao, d_aoP);
  KERNEL_CHECK(h);
  // A_PP = (aoP aoP^T)^2, Cholesky (lower in row-major == upper in column-major view; we just
  // hand rocSOLVER the symmetric matrix and ask for the factor it stores in the "upper" triangle
  // of the column-major view, i.e. row-major lower:  A = Lr Lr^T with Lr row-major lower.)
  double* A = (double*)isdf_ws(h, "fit_APP", sizeof(double) * (size_t)P * P);
  int* info = (int*)isdf_ws(h, "fit_info", 256);
  if (!A || !info) return ISDF_ERR_HIP;
  int rc = gemm_rm(h, 'N', 'T', P, P, nao, 1.0, d_aoP, nao, d_aoP, nao, 0.0, A, P);
  if (rc) return rc;
  hipLaunchKernelGGL(square_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)P), dim3(256), 0, h->stream, A,
                     (int64_t)P, (int64_t)P, (int64_t)P);
  KERNEL_CHECK(h);
  BLAS_TRY(h, rocsolver_dpotrf(h->blas, rocblas_fill_upper, P, A, P, info));
  int h_info = 0;
  HIP_TRY(h, hipMemcpyAsync(&h_info, info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h_info != 0)
    return isdf_fail(h, ISDF_ERR_NUM,
                     "A_PP = (aoP aoP^T)^2 is not numerically positive definite (leading minor %d of %d)",
                     h_info, P);
  // Column-major view: A_cm = U^T U with U upper (column-major).  In row-major terms U_cm = Lr^T
  // where A = Lr Lr^T.  We need Theta = A^-1 B with B = (aoP ao)^2 (P x G row-major).
  // Column-major view of B is B^T (G x P, ld = ldt):  X_cm = B_cm A^-1 = B_cm U^-1 U^-T.
  // B in grid chunks (keeps the GEMM well shaped and bounds nothing extra: written straight into theta).
  rc = gemm_rm(h, 'N', 'N', P, ngrids, nao, 1.0, d_aoP, nao, d_ao, ld, 0.0, d_theta, ldt);
  if (rc) return rc;
  hipLaunchKernelGGL(square_kernel, dim3((unsigned)cdiv(ngrids, 256), (unsigned)P), dim3(256), 0, h->stream,
                     d_theta, (int64_t)P, ngrids, ldt);
  KERNEL_CHECK(h);
  const double one = 1.0;
  BLAS_TRY(h, rocblas_dtrsm(h->blas, rocblas_side_right, rocblas_fill_upper, rocblas_operation_none,
                            rocblas_diagonal_non_unit, (rocblas_int)ngrids, P, &one, A, P, d_theta,
                            (rocblas_int)ldt));
  BLAS_TRY(h, rocblas_dtrsm(h->blas, rocblas_side_right, rocblas_fill_upper, rocblas_operation_transpose,
                            rocblas_diagonal_non_unit, (rocblas_int)ngrids, P, &one, A, P, d_theta,
                            (rocblas_int)ldt));
  return ISDF_OK;
}

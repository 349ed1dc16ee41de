// S3: least-squares fit of the interpolation vectors Theta.
//   (a) from the selection's own Cholesky rows:  Theta = T^-1 L,  T = L[:, piv]  (upper triangular)
//   (b) for an arbitrary point set:  Theta = [(aoP aoP^T)^2]^-1 (aoP ao)^2  by Cholesky
// The triangular solves run over all G right-hand sides in place on the (P, G) row-major array.
#include "common.h"
#include <cstdlib>

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

// x = x^2 + y^2 (elementwise), x (rows, ld), y (rows, ldy)
__global__ void square_add_kernel(double* __restrict__ x, int64_t ld, const double* __restrict__ y, int64_t ldy,
                                  int64_t cols) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c >= cols) return;
  const double a = x[r * ld + c], b = y[r * ldy + c];
  x[r * ld + c] = fma(b, b, a * a);
}

// rot[p, m] = (m < nh) ? aoP[p, nh + m] : -aoP[p, m - nh]     (aoP is (P, 2 nh))
__global__ void rotate_kernel(const double* __restrict__ aoP, int P, int nh, double* __restrict__ rot) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (m >= 2 * nh) return;
  rot[(int64_t)p * 2 * nh + m] = (m < nh) ? aoP[(int64_t)p * 2 * nh + nh + m] : -aoP[(int64_t)p * 2 * nh + m - nh];
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

// rows <- aoP ao (P x ng), element-wise squared when `square`: the own MFMA NN kernel with the square in its epilogue where the
// operands fit it (gemm_f64.hip), else rocBLAS and a separate pass
static int product_rows(isdf_handle h, int P, int64_t ng, int nao, const double* d_aoP, const double* d_ao, int64_t ld,
                        double* d_B, int64_t ldb, bool square) {
  if (gemm_nn_f64_supported(h, P, ng, nao, d_aoP, nao, d_ao, ld))
    return gemm_nn_f64(h, P, ng, nao, d_aoP, nao, d_ao, ld, d_B, ldb, square);
  int rc = gemm_rm(h, 'N', 'N', P, ng, nao, 1.0, d_aoP, nao, d_ao, ld, 0.0, d_B, ldb);
  if (rc || !square) return rc;
  hipLaunchKernelGGL(square_kernel, dim3((unsigned)cdiv(ng, 256), (unsigned)P), dim3(256), 0, h->stream, d_B, (int64_t)P, ng,
                     ldb);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

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

namespace {
__global__ void add_diag_kernel(double* __restrict__ A, int n, const double* __restrict__ maxdiag, double rel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[(int64_t)i * n + i] += rel * maxdiag[0];
}
__global__ void max_diag_kernel(const double* __restrict__ A, int n, double* __restrict__ out) {
  // single workgroup
  __shared__ double red[256];
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmax(m, A[(int64_t)i * n + i]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}
}  // namespace

extern "C" int isdf_fit_prepare(isdf_handle h, const double* d_ao, int nao, int64_t ld,
                                const int64_t* d_ip, int P, double reg_rel, double* d_aoP,
                                double* d_chol, double* reg_used) {
  return isdf_fit_prepare_cplx(h, d_ao, nao, 0, ld, d_ip, P, reg_rel, d_aoP, d_chol, reg_used);
}

extern "C" int isdf_fit_prepare_cplx(isdf_handle h, const double* d_ao, int nao, int nh, int64_t ld,
                                     const int64_t* d_ip, int P, double reg_rel, double* d_aoP,
                                     double* d_chol, double* reg_used) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, nh == 0 || 2 * nh == nao);
  ARG_CHECK(h, d_ao && d_ip && d_aoP && d_chol && nao > 0 && P > 0 && reg_rel >= 0.0);
  ARG_CHECK(h, nao <= 65535 && P <= 65535);
  hipLaunchKernelGGL(transpose_gather_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)nao), dim3(256), 0,
                     h->stream, d_ao, ld, d_ip, P, nao, d_aoP);
  KERNEL_CHECK(h);
  int* info = (int*)isdf_ws(h, "fit_info", 256);
  if (!info) return ISDF_ERR_HIP;
  double* maxdiag = (double*)(info + 16);
  double reg = reg_rel;
  for (int attempt = 0; attempt < 5; ++attempt) {
    // A_PP = (aoP aoP^T)^2 (+ reg * max diag * I), Cholesky A = U^T U in the column-major view
    int rc = gemm_rm(h, 'N', 'T', P, P, nao, 1.0, d_aoP, nao, d_aoP, nao, 0.0, d_chol, P);
    if (rc) return rc;
    if (nh > 0) {
      // complex mode: A = (Re S)^2 + (Im S)^2, Im S = aoP_rot aoP^T
      double* rot = (double*)isdf_ws(h, "fit_rot", sizeof(double) * (size_t)P * nao);
      double* tmp = (double*)isdf_ws(h, "fit_tmp", sizeof(double) * (size_t)P * P);
      if (!rot || !tmp) return ISDF_ERR_HIP;
      hipLaunchKernelGGL(rotate_kernel, dim3((unsigned)cdiv(nao, 256), (unsigned)P), dim3(256), 0, h->stream, d_aoP, P, nh, rot);
      rc = gemm_rm(h, 'N', 'T', P, P, nao, 1.0, rot, nao, d_aoP, nao, 0.0, tmp, P);
      if (rc) return rc;
      hipLaunchKernelGGL(square_add_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)P), dim3(256), 0, h->stream,
                         d_chol, (int64_t)P, tmp, (int64_t)P, (int64_t)P);
    } else {
      hipLaunchKernelGGL(square_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)P), dim3(256), 0, h->stream,
                         d_chol, (int64_t)P, (int64_t)P, (int64_t)P);
    }
    if (reg > 0.0) {
      hipLaunchKernelGGL(max_diag_kernel, dim3(1), dim3(256), 0, h->stream, d_chol, P, maxdiag);
      hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, h->stream, d_chol, P,
                         maxdiag, reg);
    }
    KERNEL_CHECK(h);
    { ProfScope ps(h, "rocsolver_dpotrf[flop]", (double)P * P * P / 3.0);
    BLAS_TRY(h, rocsolver_dpotrf(h->blas, rocblas_fill_upper, P, d_chol, P, info)); }
    int h_info = 0;
    HIP_TRY(h, hipMemcpyAsync(&h_info, info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h_info == 0) {
      if (reg_used) *reg_used = reg;
      return ISDF_OK;
    }
    reg = (reg > 0.0) ? reg * 100.0 : 1e-14;
  }
  return isdf_fail(h, ISDF_ERR_NUM,
                   "A_PP = (aoP aoP^T)^2 is not numerically positive definite even with diagonal shift %g * max diag",
                   reg / 100.0);
}

extern "C" int isdf_fit_apply(isdf_handle h, const double* d_chol, const double* d_aoP, int P, int nao,
                              const double* d_ao, int64_t ng, int64_t ld, int forward_only,
                              double* d_theta, int64_t ldt) {
  return isdf_fit_apply_cplx(h, d_chol, d_aoP, P, nao, 0, d_ao, ng, ld, forward_only, d_theta, ldt);
}

extern "C" int isdf_fit_apply_cplx(isdf_handle h, const double* d_chol, const double* d_aoP, int P, int nao,
                                   int nh, const double* d_ao, int64_t ng, int64_t ld, int forward_only,
                                   double* d_theta, int64_t ldt) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, nh == 0 || 2 * nh == nao);
  ARG_CHECK(h, d_chol && d_aoP && d_ao && d_theta && P > 0 && nao > 0 && ng > 0 && ld >= ng && ldt >= ng);
  ARG_CHECK(h, P <= 65535 && ldt < (int64_t)2147483647 && ng < (int64_t)2147483647);
  // B = (aoP ao)^2 (P x ng, row-major) written straight into theta; then in the column-major view
  // X = B_cm U^-1 U^-T  (A = U^T U).
  int rc = product_rows(h, P, ng, nao, d_aoP, d_ao, ld, d_theta, ldt, nh == 0);
  if (rc) return rc;
  if (nh > 0) {
    // complex mode: B = (aoP X)^2 + (aoP_rot X)^2, the second product in column chunks
    const int64_t CH = 8192;
    double* rot = (double*)isdf_ws(h, "fit_rot", sizeof(double) * (size_t)P * nao);
    double* tmp = (double*)isdf_ws(h, "fit_tmpB", sizeof(double) * (size_t)P * CH);
    if (!rot || !tmp) return ISDF_ERR_HIP;
    hipLaunchKernelGGL(rotate_kernel, dim3((unsigned)cdiv(nao, 256), (unsigned)P), dim3(256), 0, h->stream, d_aoP, P, nh, rot);
    for (int64_t c0 = 0; c0 < ng; c0 += CH) {
      const int64_t nc = std::min(CH, ng - c0);
      rc = gemm_rm(h, 'N', 'N', P, nc, nao, 1.0, rot, nao, d_ao + c0, ld, 0.0, tmp, CH);
      if (rc) return rc;
      hipLaunchKernelGGL(square_add_kernel, dim3((unsigned)cdiv(nc, 256), (unsigned)P), dim3(256), 0, h->stream,
                         d_theta + c0, ldt, tmp, CH, nc);
    }
  }
  KERNEL_CHECK(h);
  // Theta = A^-1 B with A = Lr Lr^T (Lr = the factor read row-major, lower): forward solve Y = Lr^-1 B, then (unless the
  // caller wants Y, see isdf_W_from_factor) the backward solve Theta = Lr^-T Y; blocked left-looking (trsm.hip: tri_left)
  rc = tri_left(h, false, P, ng, d_chol, P, d_theta, ldt);
  if (rc || forward_only) return rc;
  return tri_left(h, true, P, ng, d_chol, P, d_theta, ldt);
}

extern "C" int isdf_fit_global(isdf_handle h, const double* d_ao, int nao, int64_t ngrids, int64_t ld,
                               const int64_t* d_ip, int P, double reg_rel, double* d_theta,
                               int64_t ldt, double* d_aoP, double* reg_used) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, P > 0 && ngrids >= P);
  double* A = (double*)isdf_ws(h, "fit_APP", sizeof(double) * (size_t)P * P);
  if (!A) return ISDF_ERR_HIP;
  int rc = isdf_fit_prepare(h, d_ao, nao, ld, d_ip, P, reg_rel, d_aoP, A, reg_used);
  if (rc) return rc;
  return isdf_fit_apply(h, A, d_aoP, P, nao, d_ao, ngrids, ld, 0, d_theta, ldt);
}

extern "C" int isdf_gather_T(isdf_handle h, const double* d_L, int k, int64_t ldL, const int64_t* d_piv,
                             double* d_T) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_L && d_piv && d_T && k > 0 && k <= 65535);
  hipLaunchKernelGGL(gather_T_kernel, dim3((unsigned)cdiv(k, 256), (unsigned)k), dim3(256), 0, h->stream, d_L,
                     ldL, d_piv, k, d_T);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_W_from_factor(isdf_handle h, const double* d_F, int P, int kind, double* d_M,
                                  int64_t ldm) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_F && d_M && P > 0 && ldm >= P && (kind == 0 || kind == 1 || kind == 2));
  const double one = 1.0;
  // kinds 0 and 2: F is the column-major upper U of A = U^T U, i.e. the row-major lower L with A = L L^T
  if (kind == 2) {
    // M <- U^-T M U^-1 = L^-1 M L^-T  (the other half of A^-1 M A^-1; follow with kind 0)
    int rc = tri_left(h, false, P, P, d_F, P, d_M, ldm);
    if (rc) return rc;
    return tri_right(h, true, P, P, d_F, P, d_M, ldm);
  } else if (kind == 0) {
    // Theta = U^-1 Y  =>  W = U^-1 M U^-T = L^-T M L^-1
    int rc = tri_left(h, true, P, P, d_F, P, d_M, ldm);
    if (rc) return rc;
    return tri_right(h, false, P, P, d_F, P, d_M, ldm);
  } else {
    // F = T (row-major upper) from isdf_gather_T, i.e. column-major lower Lc = T^T,
    // Theta = T^-1 L  =>  W = T^-1 M T^-T = Lc^-T M Lc^-1   (global selection only: rocBLAS)
    ProfScope ps(h, "rocblas_dtrsm[flop]", 2.0 * (double)P * P * P, 2);
    BLAS_TRY(h, rocblas_dtrsm(h->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                              rocblas_diagonal_non_unit, P, P, &one, d_F, P, d_M, (rocblas_int)ldm));
    BLAS_TRY(h, rocblas_dtrsm(h->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none,
                              rocblas_diagonal_non_unit, P, P, &one, d_F, P, d_M, (rocblas_int)ldm));
  }
  return ISDF_OK;
}


// ---- block-Jacobi route: no triangular solve over the grid at all ------------------------------------
// With A = A_PP, B = A_P and D = blockdiag(chol(A_bb)) over the per-atom point blocks:
//   Y' = D^-1 B (cheap, block by block),  M' = w conv(Y') Y'^T,  A' = D^-1 A D^-T,
//   W  = A^-1 [w conv(B) B^T] A^-1 = D^-T [A'^-1 M' A'^-1] D^-1.
// Same W as the Cholesky routes in exact arithmetic; the block scaling keeps the small components of
// M' resolved and lowers the condition number that the two-sided inverse squares.

extern "C" int isdf_gram_sq(isdf_handle h, const double* d_aoP, int P, int nao, int nh, double* d_A) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_aoP && d_A && P > 0 && nao > 0 && P <= 65535 && (nh == 0 || 2 * nh == nao));
  int rc = gemm_rm(h, 'N', 'T', P, P, nao, 1.0, d_aoP, nao, d_aoP, nao, 0.0, d_A, P);
  if (rc) return rc;
  if (nh > 0) {
    double* rot = (double*)isdf_ws(h, "fit_rot", sizeof(double) * (size_t)P * nao);
    double* tmp = (double*)isdf_ws(h, "fit_tmp", sizeof(double) * (size_t)P * P);
    if (!rot || !tmp) return ISDF_ERR_HIP;
    hipLaunchKernelGGL(rotate_kernel, dim3((unsigned)cdiv(nao, 256), (unsigned)P), dim3(256), 0, h->stream, d_aoP, P, nh, rot);
    rc = gemm_rm(h, 'N', 'T', P, P, nao, 1.0, rot, nao, d_aoP, nao, 0.0, tmp, P);
    if (rc) return rc;
    hipLaunchKernelGGL(square_add_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)P), dim3(256), 0, h->stream, d_A,
                       (int64_t)P, tmp, (int64_t)P, (int64_t)P);
  } else {
    hipLaunchKernelGGL(square_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)P), dim3(256), 0, h->stream, d_A,
                       (int64_t)P, (int64_t)P, (int64_t)P);
  }
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_pair_gram_rows(isdf_handle h, const double* d_aoP, int P, int nao, int nh, const double* d_ao,
                                   int64_t ng, int64_t ld, double* d_B, int64_t ldb) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_aoP && d_ao && d_B && P > 0 && P <= 65535 && nao > 0 && ng > 0 && ld >= ng && ldb >= ng);
  ARG_CHECK(h, nh == 0 || 2 * nh == nao);
  int rc = product_rows(h, P, ng, nao, d_aoP, d_ao, ld, d_B, ldb, nh == 0);
  if (rc) return rc;
  if (nh > 0) {
    const int64_t CH = 8192;
    double* rot = (double*)isdf_ws(h, "fit_rot", sizeof(double) * (size_t)P * nao);
    double* tmp = (double*)isdf_ws(h, "fit_tmpB", sizeof(double) * (size_t)P * CH);
    if (!rot || !tmp) return ISDF_ERR_HIP;
    hipLaunchKernelGGL(rotate_kernel, dim3((unsigned)cdiv(nao, 256), (unsigned)P), dim3(256), 0, h->stream, d_aoP, P, nh, rot);
    for (int64_t c0 = 0; c0 < ng; c0 += CH) {
      const int64_t nc = std::min(CH, ng - c0);
      rc = gemm_rm(h, 'N', 'N', P, nc, nao, 1.0, rot, nao, d_ao + c0, ld, 0.0, tmp, CH);
      if (rc) return rc;
      hipLaunchKernelGGL(square_add_kernel, dim3((unsigned)cdiv(nc, 256), (unsigned)P), dim3(256), 0, h->stream,
                         d_B + c0, ldb, tmp, CH, nc);
    }
  }
  KERNEL_CHECK(h);
  return ISDF_OK;
}

namespace {
__global__ void copy_block_kernel(const double* __restrict__ A, int P, int off, int nb, double* __restrict__ D) {
  // D (P x P, zero elsewhere) diagonal block <- A diagonal block
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= nb) return;
  D[(int64_t)(off + r) * P + off + c] = A[(int64_t)(off + r) * P + off + c];
}
// F (n <= 8 rows, ng) = E (n, P) * Y (P, ng): one grid column per lane, the P-long dot products streamed row by row
// (every load of a wave is one coalesced 512-byte row segment); E staged through LDS in 64-row pieces.  HBM-bound:
// reads Y once.  (rocBLAS dgemm with M = 8 moves the same bytes at under 1 TB/s.)
__global__ __launch_bounds__(256) void skinny_rows_kernel(const double* __restrict__ E, int n, int P, int64_t ldE,
                                                          const double* __restrict__ Y, int64_t ldy, int64_t ng,
                                                          double* __restrict__ F, int64_t ldf, int accumulate) {
  __shared__ double sE[8][64];
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = g < ng;
  double acc[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
  for (int p0 = 0; p0 < P; p0 += 64) {
    const int np = min(64, P - p0);
    __syncthreads();
    for (int t = threadIdx.x; t < 8 * 64; t += 256) {
      const int j = t >> 6, pp = t & 63;
      sE[j][pp] = (j < n && pp < np) ? E[(int64_t)j * ldE + p0 + pp] : 0.0;
    }
    __syncthreads();
    if (live) {
      const double* y = Y + (int64_t)p0 * ldy + g;
#pragma unroll 8
      for (int pp = 0; pp < np; ++pp) {
        const double v = y[(int64_t)pp * ldy];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fma(sE[j][pp], v, acc[j]);
      }
    }
  }
  if (live)
    for (int j = 0; j < n; ++j) F[(int64_t)j * ldf + g] = accumulate ? F[(int64_t)j * ldf + g] + acc[j] : acc[j];
}

__global__ void add_diag_const_kernel(double* __restrict__ A, int n, int64_t ld, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[(int64_t)i * ld + i] += v;
}
}  // namespace

extern "C" int isdf_block_chol(isdf_handle h, const double* d_A, int P, int nblk, const int32_t* blk_off,
                               double shift_rel, double* d_D, double* shift_used) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_A && d_D && blk_off && P > 0 && nblk > 0 && blk_off[0] == 0 && blk_off[nblk] == P && shift_rel >= 0);
  int* info = (int*)isdf_ws(h, "fit_info", 256);
  if (!info) return ISDF_ERR_HIP;
  double* maxdiag = (double*)(info + 16);
  HIP_TRY(h, hipMemsetAsync(d_D, 0, sizeof(double) * (size_t)P * P, h->stream));
  hipLaunchKernelGGL(max_diag_kernel, dim3(1), dim3(256), 0, h->stream, d_A, P, maxdiag);
  double md = 0.0;
  HIP_TRY(h, hipMemcpyAsync(&md, maxdiag, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  double worst = 0.0;
  for (int b = 0; b < nblk; ++b) {
    const int off = blk_off[b], nb = blk_off[b + 1] - off;
    if (nb <= 0) continue;
    ARG_CHECK(h, nb <= 65535);
    // D is only a preconditioner: any invertible block works as long as the same D is used throughout, so a block
    // that is not numerically positive definite (its points were selected on slightly different AO values, or the
    // set is over-complete) simply gets a larger diagonal shift: shift_rel, then 1e-14, x100 per retry
    double reg = shift_rel;
    int h_info = 1;
    for (int attempt = 0; attempt < 6 && h_info != 0; ++attempt) {
      hipLaunchKernelGGL(copy_block_kernel, dim3((unsigned)cdiv(nb, 128), (unsigned)nb), dim3(128), 0, h->stream, d_A, P,
                         off, nb, d_D);
      if (reg > 0)
        hipLaunchKernelGGL(add_diag_const_kernel, dim3((unsigned)cdiv(nb, 256)), dim3(256), 0, h->stream,
                           d_D + (int64_t)off * P + off, nb, (int64_t)P, reg * md);
      KERNEL_CHECK(h);
      // column-major upper factor of the block == row-major lower D_b (A_bb = D_b D_b^T)
      BLAS_TRY(h, rocsolver_dpotrf(h->blas, rocblas_fill_upper, nb, d_D + (int64_t)off * P + off, P, info));
      HIP_TRY(h, hipMemcpyAsync(&h_info, info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      if (h_info == 0) { worst = std::max(worst, reg); break; }
      reg = (reg > 0.0) ? reg * 100.0 : 1e-14;
    }
    if (h_info != 0)
      return isdf_fail(h, ISDF_ERR_NUM, "diagonal block %d of A_PP is not positive definite even with shift %g * max diag "
                       "(minor %d of %d)", b, reg / 100.0, h_info, nb);
  }
  if (shift_used) *shift_used = worst;
  return ISDF_OK;
}

extern "C" int isdf_block_solve(isdf_handle h, const double* d_D, int P, int nblk, const int32_t* blk_off, int side,
                                int trans, double* d_X, int64_t n, int64_t ldx) {
  // side 0: rows   X (P, n)  <- op(D)^-1 X          (per block of rows)
  // side 1: cols   X (n, P)  <- X op(D)^-1          (per block of columns)
  // op(D) = D (trans 0) or D^T (trans 1); D_b row-major lower = column-major upper U_b = D_b^T
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_D && d_X && blk_off && P > 0 && nblk > 0 && n > 0 && (side == 0 || side == 1) && (trans == 0 || trans == 1));
  ARG_CHECK(h, n < 2147483647LL && ldx < 2147483647LL);
  if (side == 0 && trans == 0 && !h->trsm_substitution) {
    // the hot one (Y' = D^-1 B over the whole grid): all blocks in one launch of the hand-written kernel
    ARG_CHECK(h, blk_off[0] == 0 && blk_off[nblk] <= P);
    return block_forward_solve(h, d_D, P, nblk, blk_off, d_X, ldx, n);
  }
  for (int b = 0; b < nblk; ++b) {
    const int off = blk_off[b], nb = blk_off[b + 1] - off;
    if (nb <= 0) continue;
    const double* Db = d_D + (int64_t)off * P + off;     // row-major lower D_b, leading dimension P
    const int rc = (side == 0) ? tri_left(h, trans != 0, nb, n, Db, P, d_X + (int64_t)off * ldx, ldx)
                               : tri_right(h, trans != 0, nb, n, Db, P, d_X + off, ldx);
    if (rc) return rc;
  }
  return ISDF_OK;
}

namespace {
__global__ void block_identity_kernel(double* __restrict__ E, int P, const int32_t* __restrict__ blk_of_row) {
  // E (P x P) <- 0 with ones on the diagonal (the right-hand side whose block solves are the block inverses)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)P * P) E[i] = (i / P == i % P) ? 1.0 : 0.0;
  (void)blk_of_row;
}
}  // namespace

extern "C" int isdf_block_invert(isdf_handle h, const double* d_D, int P, int nblk, const int32_t* blk_off, double* d_Dinv) {
  // Dinv <- blockdiag(D_b^-1): forward solves of the block factors against the identity.  Entries above the diagonal and
  // outside the blocks are exact zeros (forward substitution of a unit vector never touches the rows above it).
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_D && d_Dinv && blk_off && P > 0 && nblk > 0 && blk_off[0] == 0 && blk_off[nblk] <= P);
  hipLaunchKernelGGL(block_identity_kernel, dim3((unsigned)cdiv((int64_t)P * P, 256)), dim3(256), 0, h->stream, d_Dinv, P,
                     (const int32_t*)nullptr);
  KERNEL_CHECK(h);
  return block_forward_solve(h, d_D, P, nblk, blk_off, d_Dinv, P, P);
}

extern "C" int isdf_block_apply(isdf_handle h, const double* d_Dinv, int64_t ldd, int nblk, const int32_t* blk_off,
                                double* d_X, int64_t n, int64_t ldx) {
  // X (rows of the blocks, n columns) <- Dinv_b X_b, the MFMA form of isdf_block_solve(side 0, trans 0)
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_Dinv && d_X && blk_off && nblk > 0 && n > 0 && ldx >= n && blk_off[0] == 0 && blk_off[nblk] <= ldd);
  return block_apply_inverse(h, d_Dinv, ldd, nblk, blk_off, d_X, ldx, n);
}

extern "C" int isdf_pair_rows_block_apply(isdf_handle h, const double* d_aoP, int P, int nao, const double* d_ao, int64_t ng,
                                          int64_t ld, const double* d_Dinv, int64_t ldd, int nblk, const int32_t* blk_off,
                                          double* d_B, int64_t ldb) {
  // B (P rows = the rows of the blocks, ng) <- Dinv_b (aoP ao)^2: the pair-gram rows and the block solves of the S3c route
  // in two passes over B instead of three - the product aoP ao (rocBLAS; the own NN kernel under option gemm_nn_own, which
  // squares in its epilogue), then the MFMA block apply, which otherwise squares its input while staging it
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_aoP && d_ao && d_Dinv && d_B && blk_off && P > 0 && nao > 0 && ng > 0 && ld >= ng && ldb >= ng && nblk > 0);
  ARG_CHECK(h, blk_off[0] == 0 && blk_off[nblk] == P);
  const bool own = gemm_nn_f64_supported(h, P, ng, nao, d_aoP, nao, d_ao, ld);   // squares in its epilogue
  int rc = product_rows(h, P, ng, nao, d_aoP, d_ao, ld, d_B, ldb, own);
  if (rc) return rc;
  return block_apply_inverse(h, d_Dinv, ldd, nblk, blk_off, d_B, ldb, ng, !own);
}

extern "C" int isdf_shift_diag(isdf_handle h, double* d_A, int P, double shift_rel) {
  // A <- A + shift_rel * max(diag A) * I  (the fit's regularisation, applied before the block scaling of S3c so that
  // both fit routes solve the same regularised normal equations)
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_A && P > 0 && shift_rel >= 0);
  if (shift_rel == 0.0) return ISDF_OK;
  int* info = (int*)isdf_ws(h, "fit_info", 256);
  if (!info) return ISDF_ERR_HIP;
  double* maxdiag = (double*)(info + 16);
  hipLaunchKernelGGL(max_diag_kernel, dim3(1), dim3(256), 0, h->stream, d_A, P, maxdiag);
  hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, h->stream, d_A, P, maxdiag, shift_rel);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_chol_inplace(isdf_handle h, double* d_A, int P, double shift_rel, double* d_scratch,
                                 double* reg_used) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_A && P > 0 && shift_rel >= 0);
  int* info = (int*)isdf_ws(h, "fit_info", 256);
  // the unshifted matrix is kept for the retries: in the caller's scratch (P*P doubles) when given
  double* keep = d_scratch ? d_scratch : (double*)isdf_ws(h, "chol_keep", sizeof(double) * (size_t)P * P);
  if (!info || !keep) return ISDF_ERR_HIP;
  double* maxdiag = (double*)(info + 16);
  HIP_TRY(h, hipMemcpyAsync(keep, d_A, sizeof(double) * (size_t)P * P, hipMemcpyDeviceToDevice, h->stream));
  // same ladder as isdf_fit_prepare: an over-complete point set makes the matrix numerically singular
  double reg = shift_rel;
  for (int attempt = 0; attempt < 5; ++attempt) {
    if (attempt > 0)
      HIP_TRY(h, hipMemcpyAsync(d_A, keep, sizeof(double) * (size_t)P * P, hipMemcpyDeviceToDevice, h->stream));
    if (reg > 0) {
      hipLaunchKernelGGL(max_diag_kernel, dim3(1), dim3(256), 0, h->stream, d_A, P, maxdiag);
      hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, h->stream, d_A, P, maxdiag, reg);
      KERNEL_CHECK(h);
    }
    { ProfScope ps(h, "rocsolver_dpotrf[flop]", (double)P * P * P / 3.0);
      BLAS_TRY(h, rocsolver_dpotrf(h->blas, rocblas_fill_upper, P, d_A, P, info)); }
    int h_info = 0;
    HIP_TRY(h, hipMemcpyAsync(&h_info, info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h_info == 0) {
      if (reg_used) *reg_used = reg;
      return ISDF_OK;
    }
    reg = (reg > 0.0) ? reg * 100.0 : 1e-14;
  }
  return isdf_fail(h, ISDF_ERR_NUM, "matrix is not numerically positive definite even with diagonal shift %g * max diag",
                   reg / 100.0);
}

extern "C" int isdf_factor_solve(isdf_handle h, const double* d_fac, int P, double* d_X, int64_t n, int64_t ldx) {
  // X (P, n) row-major <- A^-1 X = L^-T L^-1 X, A = L L^T (d_fac: row-major lower L == the column-major upper U of
  // isdf_fit_prepare / isdf_chol_inplace)
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_fac && d_X && P > 0 && n > 0 && ldx >= n && n < 2147483647LL && ldx < 2147483647LL);
  int rc = tri_left(h, false, P, n, d_fac, P, d_X, ldx);
  if (rc) return rc;
  return tri_left(h, true, P, n, d_fac, P, d_X, ldx);
}

extern "C" int isdf_bj_probe_vectors(isdf_handle h, double* d_T, int n, const double* d_fac, const double* d_D, int P,
                                     int nblk, const int32_t* blk_off) {
  // T (n, P) rows t_j  ->  e_j = A'^-1 D^-1 t_j (in place)
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_T && d_fac && d_D && n > 0 && P > 0);
  int rc = isdf_block_solve(h, d_D, P, nblk, blk_off, 1, 1, d_T, n, P);      // t^T D^-T = (D^-1 t)^T
  if (rc) return rc;
  // rows t^T <- t^T A'^-1 = t^T L^-T L^-1  (A' = L L^T symmetric)
  rc = tri_right(h, true, P, n, d_fac, P, d_T, P);
  if (rc) return rc;
  return tri_right(h, false, P, n, d_fac, P, d_T, P);
}

extern "C" int isdf_rows_combine(isdf_handle h, const double* d_E, int n, int64_t ldE, int rows, const double* d_Y,
                                 int64_t ng, int64_t ldy, double* d_F, int64_t ldf, int accumulate) {
  // F (n, ng) (+)= E (n, rows; leading dimension ldE) Y (rows, ng): a few combinations of many long rows, one pass over Y
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_E && d_Y && d_F && n > 0 && rows > 0 && ng > 0 && ldE >= rows && ldy >= ng && ldf >= ng);
  if (n > 8) return gemm_rm(h, 'N', 'N', n, ng, rows, 1.0, d_E, ldE, d_Y, ldy, accumulate ? 1.0 : 0.0, d_F, ldf);
  ProfScope ps(h, "skinny_rows_kernel[byte]", 8.0 * (double)rows * (double)ng);
  hipLaunchKernelGGL(skinny_rows_kernel, dim3((unsigned)cdiv(ng, 256)), dim3(256), 0, h->stream, d_E, n, rows, ldE, d_Y, ldy,
                     ng, d_F, ldf, accumulate ? 1 : 0);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_bj_probe_rows(isdf_handle h, double* d_T, int n, const double* d_fac, const double* d_D, int P,
                                  int nblk, const int32_t* blk_off, const double* d_Yp, int64_t ng, int64_t ldy,
                                  double* d_F, int64_t ldf) {
  // T (n, P) rows t_j  ->  e_j = A'^-1 D^-1 t_j (in place),  F (n, ng) = E Y' = (Theta^T t_j) on the grid columns
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_T && d_fac && d_D && d_Yp && d_F && n > 0 && P > 0 && ng > 0 && ldy >= ng && ldf >= ng);
  int rc = isdf_bj_probe_vectors(h, d_T, n, d_fac, d_D, P, nblk, blk_off);
  if (rc) return rc;
  return isdf_rows_combine(h, d_T, n, P, P, d_Yp, ng, ldy, d_F, ldf, 0);
}

// ---- (AO x occupied orbital) pair space ------------------------------------------------------------------------
// The exchange of a density D = sum_i psi_i psi_i^T only needs the pair products phi_mu psi_i (the reduction the reference makes
// with mo_coeff-tagged density matrices, pyscf/pbc/df/fft_jk.py:206-210,235-238).  Their Gram matrix is
//   A(r, r') = (sum_mu phi_mu(r) phi_mu(r')) * (sum_i psi_i(r) psi_i(r'))
// - the element-wise PRODUCT of two Gram matrices where the AO x AO pair space has the square of one.  Selection, fit, W and K
// keep their formulas; only these two products change.
namespace {
// x (rows, ld) .*= y (rows, ldy) on `cols` columns
__global__ void mul_rows_kernel(double* __restrict__ x, int64_t ld, const double* __restrict__ y, int64_t ldy, int64_t cols) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c >= cols) return;
  x[r * ld + c] *= y[r * ldy + c];
}
// columns per chunk of the second factor: its (P x chunk) scratch stays under 2 GiB
inline int64_t prod_chunk(int P) {
  int64_t ch = ((int64_t)1 << 28) / std::max(P, 1);
  ch = std::max<int64_t>(1024, std::min<int64_t>(65536, ch));
  return ch / 256 * 256;
}
}  // namespace

extern "C" int isdf_gram_prod(isdf_handle h, const double* d_aoP, int P, int nao, const double* d_psiP, int nocc, double* d_A) {
  // A (P, P) = (aoP aoP^T) o (psiP psiP^T)
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_aoP && d_psiP && d_A && P > 0 && P <= 65535 && nao > 0 && nocc > 0);
  int rc = gemm_rm(h, 'N', 'T', P, P, nao, 1.0, d_aoP, nao, d_aoP, nao, 0.0, d_A, P);
  if (rc) return rc;
  const int64_t CH = prod_chunk(P);                          // rows of the second factor per pass
  double* tmp = (double*)isdf_ws(h, "fit_tmpB", sizeof(double) * (size_t)P * CH);
  if (!tmp) return ISDF_ERR_HIP;
  for (int64_t r0 = 0; r0 < P; r0 += CH) {
    const int64_t nr = std::min<int64_t>(CH, P - r0);
    rc = gemm_rm(h, 'N', 'T', nr, P, nocc, 1.0, d_psiP + r0 * nocc, nocc, d_psiP, nocc, 0.0, tmp, P);
    if (rc) return rc;
    hipLaunchKernelGGL(mul_rows_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)nr), dim3(256), 0, h->stream, d_A + r0 * P,
                       (int64_t)P, tmp, (int64_t)P, (int64_t)P);
  }
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_pair_prod_rows(isdf_handle h, const double* d_aoP, int P, int nao, const double* d_psiP, int nocc,
                                   const double* d_ao, int64_t ld, const double* d_psi, int64_t ldpsi, int64_t ng,
                                   double* d_B, int64_t ldb) {
  // B (P, ng) = (aoP ao) o (psiP psi): the second product in column chunks through the library workspace
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_aoP && d_psiP && d_ao && d_psi && d_B && P > 0 && P <= 65535 && nao > 0 && nocc > 0 && ng > 0);
  ARG_CHECK(h, ld >= ng && ldpsi >= ng && ldb >= ng);
  int rc = product_rows(h, P, ng, nao, d_aoP, d_ao, ld, d_B, ldb, false);
  if (rc) return rc;
  const int64_t CH = prod_chunk(P);
  double* tmp = (double*)isdf_ws(h, "fit_tmpB", sizeof(double) * (size_t)P * CH);
  if (!tmp) return ISDF_ERR_HIP;
  for (int64_t c0 = 0; c0 < ng; c0 += CH) {
    const int64_t nc = std::min(CH, ng - c0);
    rc = gemm_rm(h, 'N', 'N', P, nc, nocc, 1.0, d_psiP, nocc, d_psi + c0, ldpsi, 0.0, tmp, CH);
    if (rc) return rc;
    ProfScope ps(h, "mul_rows_kernel[byte]", 24.0 * (double)P * (double)nc);
    hipLaunchKernelGGL(mul_rows_kernel, dim3((unsigned)cdiv(nc, 256), (unsigned)P), dim3(256), 0, h->stream, d_B + c0, ldb, tmp,
                       CH, nc);
  }
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_factor_solve_half(isdf_handle h, const double* d_fac, int P, int backward, double* d_X, int64_t n,
                                      int64_t ldx) {
  // X (P, n) row-major <- L^-1 X (backward 0) or L^-T X (backward 1), A = L L^T (d_fac as in isdf_factor_solve): the two halves
  // of the Cholesky fit route for rows the caller produced itself (isdf_pair_prod_rows)
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_fac && d_X && P > 0 && n > 0 && ldx >= n && n < 2147483647LL && ldx < 2147483647LL);
  return tri_left(h, backward != 0, P, n, d_fac, P, d_X, ldx);
}

extern "C" int isdf_gather_aoP(isdf_handle h, const double* d_ao, int nao, int64_t ld, const int64_t* d_ip, int P,
                               double* d_aoP) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_ao && d_ip && d_aoP && nao > 0 && nao <= 65535 && P > 0);
  hipLaunchKernelGGL(transpose_gather_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)nao), dim3(256), 0, h->stream, d_ao,
                     ld, d_ip, P, nao, d_aoP);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

// S6/S7: J (exact, FFTDF formula) and K (from the interpolation factorisation) on the device.
//   J: rho_g = sum_mn dm_mn ao_mg ao_ng ;  v = (vol/G) ifft(coulG fft rho).real ; vj = ao (v .* ao)^T
//      (pyscf/pbc/df/fft_jk.py:63-107)
//   K: vk = aoP^T [ (aoP dm aoP^T) .* W ] aoP                       
#include "common.h"

namespace {

constexpr int64_t JCHUNK = 32768;   // grid columns per pass (workspace nao * JCHUNK doubles)

// rho[g] = sum_mu T[mu,g] * ao[mu,g]   (T = dm * ao for this chunk)
__global__ void rho_reduce_kernel(const double* __restrict__ T, int64_t ldT,
                                  const double* __restrict__ ao, int64_t ld, int nao, int64_t ng,
                                  double* __restrict__ rho) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  double s = 0.0;
#pragma unroll 4
  for (int mu = 0; mu < nao; ++mu) s = fma(T[(int64_t)mu * ldT + g], ao[(int64_t)mu * ld + g], s);
  rho[g] = s;
}

__global__ void hadamard_kernel(double* __restrict__ M, int64_t ldm, const double* __restrict__ W,
                                int64_t ldw, int64_t rows, int64_t cols) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c >= cols) return;
  M[r * ldm + c] *= W[r * ldw + c];
}

// rho[(i,j), g] = ao[i0+i, g] * mo[j, g]
__global__ void pair_rows_kernel(const double* __restrict__ ao, int64_t ld, const double* __restrict__ mo, int64_t ldmo,
                                 int nocc, int64_t ng, double* __restrict__ R) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const int row = blockIdx.y;               // i * nocc + j
  const int i = row / nocc, jj = row % nocc;
  R[(int64_t)row * ng + g] = ao[(int64_t)i * ld + g] * mo[(int64_t)jj * ldmo + g];
}

// vdm[i, g] = sum_j R[(i,j), g] * mo[j, g]
__global__ void pair_reduce_kernel(const double* __restrict__ R, const double* __restrict__ mo, int64_t ldmo, int nocc,
                                   int64_t ng, double* __restrict__ vdm) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const int i = blockIdx.y;
  double s = 0.0;
  for (int jj = 0; jj < nocc; ++jj) s = fma(R[((int64_t)i * nocc + jj) * ng + g], mo[(int64_t)jj * ldmo + g], s);
  vdm[(int64_t)i * ng + g] = s;
}

}  // namespace

extern "C" int isdf_rho(isdf_handle h, const double* d_ao, int nao, int64_t ng, int64_t ld,
                        const double* d_dm, int nset, double* d_rho, int64_t ldrho) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_ao && d_dm && d_rho && nao > 0 && ng > 0 && ld >= ng && nset > 0 && ldrho >= ng);
  double* T = (double*)isdf_ws(h, "jk_T", sizeof(double) * (size_t)nao * JCHUNK);
  if (!T) return ISDF_ERR_HIP;
  for (int i = 0; i < nset; ++i) {
    for (int64_t g0 = 0; g0 < ng; g0 += JCHUNK) {
      const int64_t nc = std::min(JCHUNK, ng - g0);
      int rc = gemm_rm(h, 'N', 'N', nao, nc, nao, 1.0, d_dm + (int64_t)i * nao * nao, nao, d_ao + g0, ld, 0.0, T, JCHUNK);
      if (rc) return rc;
      hipLaunchKernelGGL(rho_reduce_kernel, dim3((unsigned)cdiv(nc, 256)), dim3(256), 0, h->stream, T, JCHUNK,
                         d_ao + g0, ld, nao, nc, d_rho + (int64_t)i * ldrho + g0);
      KERNEL_CHECK(h);
    }
  }
  return ISDF_OK;
}

extern "C" int isdf_vj_from_vR(isdf_handle h, const double* d_ao, int nao, int64_t ng, int64_t ld,
                               const double* d_vR, int nset, int64_t ldv, double* d_vj) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_ao && d_vR && d_vj && nao > 0 && ng > 0 && ld >= ng && nset > 0 && ldv >= ng);
  // vj = ao (v .* ao)^T: the potential is applied to the B operand while it is staged (no scaled copy)
  for (int i = 0; i < nset; ++i) {
    int rc = gemm_nt_f64_scaled(h, nao, nao, ng, 1.0, d_ao, ld, d_ao, ld, d_vR + (int64_t)i * ldv, 0.0,
                                d_vj + (int64_t)i * nao * nao, nao);
    if (rc) return rc;
  }
  return ISDF_OK;
}

extern "C" int isdf_get_j(isdf_handle h, const double* d_ao, int nao, int64_t ngrids, int64_t ld,
                          const int32_t mesh[3], const double a[9], const double* d_dm, int nset,
                          double* d_vj) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, mesh && ngrids == (int64_t)mesh[0] * mesh[1] * mesh[2]);
  double* rho = (double*)isdf_ws(h, "jk_rho", sizeof(double) * (size_t)nset * ngrids);
  if (!rho) return ISDF_ERR_HIP;
  int rc = isdf_rho(h, d_ao, nao, ngrids, ld, d_dm, nset, rho, ngrids);
  if (rc) return rc;
  rc = isdf_coulomb_potential(h, rho, nset, ngrids, mesh, a);
  if (rc) return rc;
  return isdf_vj_from_vR(h, d_ao, nao, ngrids, ld, rho, nset, ngrids, d_vj);
}

extern "C" int isdf_get_k(isdf_handle h, const double* d_aoP, int P, int nao, const double* d_W,
                          int64_t ldw, int row0, int nrows, const double* d_dm, int nset,
                          double* d_vk) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_aoP && d_W && d_dm && d_vk && P > 0 && nao > 0 && ldw >= P && nset > 0);
  ARG_CHECK(h, row0 >= 0 && nrows >= 0 && row0 + nrows <= P);
  const int RCH = 4096;   // Hadamard rows per pass
  const int rch = std::min(RCH, std::max(nrows, 1));
  double* X = (double*)isdf_ws(h, "k_X", sizeof(double) * (size_t)P * nao);        // aoP dm^T... (P, nao)
  double* M = (double*)isdf_ws(h, "k_M", sizeof(double) * (size_t)rch * P);        // (rows, P)
  double* Y = (double*)isdf_ws(h, "k_Y", sizeof(double) * (size_t)rch * nao);      // (rows, nao)
  if (!X || !M || !Y) return ISDF_ERR_HIP;
  for (int i = 0; i < nset; ++i) {
    const double* dm = d_dm + (int64_t)i * nao * nao;
    double* vk = d_vk + (int64_t)i * nao * nao;
    // X = aoP dm^T  so that  (aoP dm aoP^T)[p,q] = sum_n (aoP dm)[p,n] aoP[q,n]; we need aoP dm:
    int rc = gemm_rm(h, 'N', 'N', P, nao, nao, 1.0, d_aoP, nao, dm, nao, 0.0, X, nao);
    if (rc) return rc;
    if (nrows == 0) {
      HIP_TRY(h, hipMemsetAsync(vk, 0, sizeof(double) * (size_t)nao * nao, h->stream));
      continue;
    }
    for (int r = row0; r < row0 + nrows; r += rch) {
      const int nr = std::min(rch, row0 + nrows - r);
      // M = X[r:r+nr] aoP^T   (nr x P)
      rc = gemm_rm(h, 'N', 'T', nr, P, nao, 1.0, X + (int64_t)r * nao, nao, d_aoP, nao, 0.0, M, P);
      if (rc) return rc;
      hipLaunchKernelGGL(hadamard_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)nr), dim3(256), 0, h->stream, M,
                         (int64_t)P, d_W + (int64_t)r * ldw, ldw, (int64_t)nr, (int64_t)P);
      KERNEL_CHECK(h);
      // Y = M aoP (nr x nao);  vk += aoP[r:r+nr]^T Y
      rc = gemm_rm(h, 'N', 'N', nr, nao, P, 1.0, M, P, d_aoP, nao, 0.0, Y, nao);
      if (rc) return rc;
      rc = gemm_rm(h, 'T', 'N', nao, nao, nr, 1.0, d_aoP + (int64_t)r * nao, nao, Y, nao, r == row0 ? 0.0 : 1.0, vk, nao);
      if (rc) return rc;
    }
  }
  return ISDF_OK;
}

extern "C" int isdf_get_k_exact(isdf_handle h, const double* d_ao, int nao, int64_t ngrids, int64_t ld,
                                const double* d_C, int nocc, const int32_t mesh[3], const double a[9],
                                int i0, int ni, int max_rows, double* d_vk) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_ao && d_C && mesh && a && d_vk && nao > 0 && nocc > 0 && max_rows >= nocc);
  ARG_CHECK(h, ngrids == (int64_t)mesh[0] * mesh[1] * mesh[2] && ld >= ngrids);
  ARG_CHECK(h, i0 >= 0 && ni >= 0 && i0 + ni <= nao && nocc <= 65535);
  if (ni == 0) return ISDF_OK;
  const int64_t G = ngrids;
  const int bi = std::max(1, std::min(ni, max_rows / nocc));        // AO rows per pass
  double* mo = (double*)isdf_ws(h, "kx_mo", sizeof(double) * (size_t)nocc * G);
  double* R = (double*)isdf_ws(h, "kx_R", sizeof(double) * (size_t)bi * nocc * G);
  double* vdm = (double*)isdf_ws(h, "kx_vdm", sizeof(double) * (size_t)bi * G);
  if (!mo || !R || !vdm) return ISDF_ERR_HIP;
  // occupied orbitals on the grid: mo = C^T ao  (C is (nao, nocc), already scaled by sqrt(occ))
  int rc = gemm_rm(h, 'T', 'N', nocc, G, nao, 1.0, d_C, nocc, d_ao, ld, 0.0, mo, G);
  if (rc) return rc;
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
                     a[2] * (a[3] * a[7] - a[4] * a[6]);
  const double w = fabs(det) / (double)G;
  for (int r = i0; r < i0 + ni; r += bi) {
    const int nb = std::min(bi, i0 + ni - r);
    ARG_CHECK(h, nb * nocc <= 65535);
    hipLaunchKernelGGL(pair_rows_kernel, dim3((unsigned)cdiv(G, 256), (unsigned)(nb * nocc)), dim3(256), 0, h->stream,
                       d_ao + (int64_t)r * ld, ld, mo, G, nocc, G, R);
    KERNEL_CHECK(h);
    rc = isdf_coulomb_rows(h, R, nb * nocc, G, mesh, a, nb * nocc, R, G);
    if (rc) return rc;
    hipLaunchKernelGGL(pair_reduce_kernel, dim3((unsigned)cdiv(G, 256), (unsigned)nb), dim3(256), 0, h->stream, R, mo, G,
                       nocc, G, vdm);
    KERNEL_CHECK(h);
    rc = gemm_nt_f64(h, nb, nao, G, w, vdm, G, d_ao, ld, 0.0, d_vk + (int64_t)r * nao, nao);
    if (rc) return rc;
  }
  return ISDF_OK;
}

// ---- robust K: building blocks ----------------------------------------------------------------------
namespace {
__global__ void hadamard_rows_kernel(double* __restrict__ X, int64_t ldx, const double* __restrict__ Y, int64_t ldy,
                                     int64_t cols) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c < cols) X[r * ldx + c] *= Y[r * ldy + c];
}
}  // namespace

extern "C" int isdf_hadamard_rows(isdf_handle h, double* d_X, int64_t ldx, const double* d_Y, int64_t ldy, int rows,
                                  int64_t cols) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_X && d_Y && rows > 0 && rows <= 65535 && cols > 0 && ldx >= cols && ldy >= cols);
  ProfScope ps(h, "hadamard_rows_kernel[byte]", 24.0 * (double)rows * (double)cols);
  hipLaunchKernelGGL(hadamard_rows_kernel, dim3((unsigned)cdiv(cols, 256), (unsigned)rows), dim3(256), 0, h->stream, d_X, ldx,
                     d_Y, ldy, cols);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_gemm_nn(isdf_handle h, int M, int64_t N, int K, double alpha, const double* d_A, int64_t lda,
                            const double* d_B, int64_t ldb, double beta, double* d_C, int64_t ldc) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_A && d_B && d_C && M > 0 && N > 0 && K > 0 && lda >= K && ldb >= N && ldc >= N);
  // the own MFMA NN kernel for the plain product on operands it fits (row panels of >= 222 rows, K % 32 == 0), rocBLAS otherwise
  if (alpha == 1.0 && beta == 0.0 && gemm_nn_f64_supported(h, M, N, K, d_A, lda, d_B, ldb))
    return gemm_nn_f64(h, M, N, K, d_A, lda, d_B, ldb, d_C, ldc, false);
  return gemm_rm(h, 'N', 'N', M, N, K, alpha, d_A, lda, d_B, ldb, beta, d_C, ldc);
}


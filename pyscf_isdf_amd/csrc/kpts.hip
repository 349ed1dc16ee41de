// k-point extension of the ISDF path (BASELINE configs[3]).
//
// Bloch AOs are held as lattice-periodic parts u^k = exp(-i k.r) phi^k in two real planes (Re rows,
// then Im rows), so selection and fit run on real arrays (select_ip.hip / fit.hip complex mode) and
// give real, k-independent interpolation vectors Theta_P(r).  Per difference vector q = k2 - k1:
//     V^q_P = ifft( coulG(q) fft(Theta_P) )            (complex: coulG(q+G) has no inversion symmetry)
//     M^q   = (vol/G) V^q Theta^T                      (Hermitian; two real MFMA GEMMs, upper half + mirror)
//     W^q   = diag(ph) S^-1 M^q S^-T diag(ph)^*,  ph_P = exp(-i q.r_P)
//     K^{k1} += 1/nk  aoP_{k1}^H [ (aoP_{k2} D^{k2} aoP_{k2}^H) .* W^q ] aoP_{k1}
// following pyscf/pbc/df/fft_jk.py:177-302 (pair density conj(ao1) exp(-i q.r) ao2, kernel
// get_coulG(cell, q), weight 1/nkpts * vol/G); J as fft_jk.py:33-109.  oracle/kisdf.py is the checker.
#include "common.h"
#include <rocblas/rocblas.h>

namespace {

__global__ void pack_real_to_complex_kernel(const double* __restrict__ in, double2* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = make_double2(in[i], 0.0);
}

__global__ void mul_full_kernel(double2* __restrict__ z, const double* __restrict__ cg, int64_t G, int64_t total,
                                double scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const double c = cg[i % G] * scale;
    double2 v = z[i];
    v.x *= c;
    v.y *= c;
    z[i] = v;
  }
}

__global__ void unpack_complex_kernel(const double2* __restrict__ in, double* __restrict__ re, double* __restrict__ im,
                                      int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double2 v = in[i];
    re[i] = v.x;
    im[i] = v.y;
  }
}

// W[q][p] = sign * W[p][q] for q > p
__global__ void mirror_upper_sign_kernel(double* __restrict__ W, int P, int64_t ldw, double sign) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x, by = blockIdx.y;
  if (bx < by) return;
  const int tx = threadIdx.x, ty = threadIdx.y;
  for (int i = ty; i < 32; i += 8) {
    const int r = by * 32 + i, cc = bx * 32 + tx;
    tile[i][tx] = (r < P && cc < P) ? W[(int64_t)r * ldw + cc] : 0.0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int r = bx * 32 + i, cc = by * 32 + tx;
    if (r < P && cc < P && r > cc) W[(int64_t)r * ldw + cc] = sign * tile[tx][i];
  }
}

// Wc[p,q] = (Wre + i Wim)[p,q] * ph[p] * conj(ph[q])
__global__ void finish_Wq_kernel(const double* __restrict__ Wre, const double* __restrict__ Wim, int P, int64_t ldw,
                                 const double2* __restrict__ ph, double2* __restrict__ Wc) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (q >= P) return;
  const double a = Wre[(int64_t)p * ldw + q];
  const double b = Wim[(int64_t)p * ldw + q];
  const double2 fp = ph[p], fq = ph[q];
  // f = ph[p] * conj(ph[q])
  const double fr = fp.x * fq.x + fp.y * fq.y;
  const double fi = fp.y * fq.x - fp.x * fq.y;
  Wc[(int64_t)p * P + q] = make_double2(a * fr - b * fi, a * fi + b * fr);
}

__global__ void zhadamard_kernel(double2* __restrict__ X, const double2* __restrict__ W, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double2 x = X[i], w = W[i];
    X[i] = make_double2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x);
  }
}

// rho[g] += scale * sum_j (Tr[j,g] ur[j,g] + Ti[j,g] ui[j,g])
__global__ void rho_k_reduce_kernel(const double* __restrict__ Tr, const double* __restrict__ Ti, int64_t ldT,
                                    const double* __restrict__ ur, const double* __restrict__ ui, int64_t ld,
                                    int nao, int64_t ng, double scale, double* __restrict__ rho) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  double s = 0.0;
  for (int j = 0; j < nao; ++j) {
    s = fma(Tr[(int64_t)j * ldT + g], ur[(int64_t)j * ld + g], s);
    s = fma(Ti[(int64_t)j * ldT + g], ui[(int64_t)j * ld + g], s);
  }
  rho[g] += scale * s;
}

// Z[(p, j), g] = conj(u1[p, g]) * m2[j, g]   (periodic parts: the Bloch phases of conj(phi^{k1}) exp(-i q.r) phi^{k2} cancel)
__global__ void pair_rows_cplx_kernel(const double* __restrict__ u1r, const double* __restrict__ u1i, int64_t ld1,
                                      const double* __restrict__ m2r, const double* __restrict__ m2i, int64_t ld2, int nocc,
                                      int64_t G, double2* __restrict__ Z) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const int row = blockIdx.y, p = row / nocc, j = row % nocc;
  const double ar = u1r[(int64_t)p * ld1 + g], ai = -u1i[(int64_t)p * ld1 + g];
  const double br = m2r[(int64_t)j * ld2 + g], bi = m2i[(int64_t)j * ld2 + g];
  Z[(int64_t)row * G + g] = make_double2(ar * br - ai * bi, ar * bi + ai * br);
}

// T[p, g] = sum_j Z[(p, j), g] * conj(m2[j, g])
__global__ void pair_reduce_cplx_kernel(const double2* __restrict__ Z, const double* __restrict__ m2r,
                                        const double* __restrict__ m2i, int64_t ld2, int nocc, int64_t G,
                                        double* __restrict__ Tr, double* __restrict__ Ti) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const int p = blockIdx.y;
  double sr = 0.0, si = 0.0;
  for (int j = 0; j < nocc; ++j) {
    const double2 z = Z[((int64_t)p * nocc + j) * G + g];
    const double br = m2r[(int64_t)j * ld2 + g], bi = m2i[(int64_t)j * ld2 + g];
    sr += z.x * br + z.y * bi;
    si += z.y * br - z.x * bi;
  }
  Tr[(int64_t)p * G + g] = sr;
  Ti[(int64_t)p * G + g] = si;
}

// (Ar + i Ai) *= (Br + i Bi), element-wise on planes of `cols` columns
__global__ void zhadamard_planes_kernel(double* __restrict__ Ar, double* __restrict__ Ai, int64_t lda, const double* __restrict__ Br,
                                        const double* __restrict__ Bi, int64_t ldb, int64_t cols) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c >= cols) return;
  const double ar = Ar[r * lda + c], ai = Ai[r * lda + c], br = Br[r * ldb + c], bi = Bi[r * ldb + c];
  Ar[r * lda + c] = ar * br - ai * bi;
  Ai[r * lda + c] = ar * bi + ai * br;
}

// A[r][a][b] = sum_t (-1)^t rows[r][...t...] along `axis` (the Nyquist component of that axis' DFT), written as complex (imag 0)
__global__ void nyquist_reduce_kernel(const double* __restrict__ rows, int64_t ld, int n0, int n1, int n2, int axis,
                                      double2* __restrict__ out) {
  const int na = axis == 0 ? n1 : n0, nb = axis == 2 ? n1 : n2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)na * nb) return;
  const int ia = (int)(idx / nb), ib = (int)(idx % nb);
  const double* r = rows + (int64_t)blockIdx.y * ld;
  const int nt = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
  int64_t base, stride;
  if (axis == 0) { base = (int64_t)ia * n2 + ib; stride = (int64_t)n1 * n2; }
  else if (axis == 1) { base = (int64_t)ia * n1 * n2 + ib; stride = n2; }
  else { base = ((int64_t)ia * n1 + ib) * n2; stride = 1; }
  double s = 0.0;
  for (int t = 0; t < nt; t += 2) s += r[base + (int64_t)t * stride] - r[base + (int64_t)(t + 1) * stride];
  out[(int64_t)blockIdx.y * na * nb + idx] = make_double2(s, 0.0);
}

inline rocblas_operation zop(char c) {
  return c == 'N' ? rocblas_operation_none : (c == 'T' ? rocblas_operation_transpose : rocblas_operation_conjugate_transpose);
}

}  // namespace

// row-major complex C (M x N) = alpha op(A) op(B) + beta C, ops in {N, T, C}
static int zgemm_rm(isdf_handle h, char opA, char opB, int M, int N, int K, double2 alpha, const double2* A, int lda,
                    const double2* B, int ldb, double2 beta, double2* C, int ldc) {
  const rocblas_double_complex al(alpha.x, alpha.y), be(beta.x, beta.y);
  ProfScope ps(h, "rocblas_zgemm[flop]", 8.0 * M * N * (double)K);
  BLAS_TRY(h, rocblas_zgemm(h->blas, zop(opB), zop(opA), N, M, K, &al, (const rocblas_double_complex*)B, ldb,
                            (const rocblas_double_complex*)A, lda, &be, (rocblas_double_complex*)C, ldc));
  return ISDF_OK;
}

static int get_z2z_plan(isdf_handle h, const int32_t mesh[3], int batch, hipfftHandle* out) {
  std::vector<int> key = {mesh[0], mesh[1], mesh[2], batch, -1};
  auto it = h->plans.find(key);
  if (it != h->plans.end()) { *out = it->second.fwd; return ISDF_OK; }
  FftPlan p;
  int dims[3] = {mesh[0], mesh[1], mesh[2]};
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  ARG_CHECK(h, G < (int64_t)2147483647);
  FFT_TRY(h, hipfftPlanMany(&p.fwd, 3, dims, nullptr, 1, (int)G, nullptr, 1, (int)G, HIPFFT_Z2Z, batch));
  FFT_TRY(h, hipfftSetStream(p.fwd, h->stream));
  p.bwd = 0;
  h->plans.emplace(key, p);
  *out = p.fwd;
  return ISDF_OK;
}

extern "C" int isdf_coulomb_Wq(isdf_handle h, const double* d_theta, int P, int64_t ldt, const int32_t mesh[3],
                               const double* d_coulG, double weight, int row0, int nrows, int batch,
                               int upper_only, double* d_Wre, double* d_Wim, int64_t ldw) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_theta && mesh && d_coulG && d_Wre && d_Wim && P > 0 && batch > 0 && ldw >= P);
  ARG_CHECK(h, row0 >= 0 && nrows >= 0 && row0 + nrows <= P);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  ARG_CHECK(h, ldt == G);
  if (nrows == 0) return ISDF_OK;
  if (batch > nrows) batch = nrows;
  double2* Z = (double2*)isdf_ws(h, "coul_Zfull", sizeof(double2) * (size_t)batch * G);
  double* Vre = (double*)isdf_ws(h, "coul_V", sizeof(double) * (size_t)batch * G);
  double* Vim = (double*)isdf_ws(h, "coul_Vim", sizeof(double) * (size_t)batch * G);
  if (!Z || !Vre || !Vim) return ISDF_ERR_HIP;
  // own FFT (fft_conv.hip: real-input forward through the half spectrum, table-expanding multiply, complex inverse) where the
  // mesh allows, hipFFT Z2Z otherwise
  const bool own = conv_rows_q_own_supported(h, mesh, batch);
  double2* Zh = nullptr;
  if (own) {
    Zh = (double2*)isdf_ws(h, "coul_Z", sizeof(double2) * (size_t)batch * mesh[0] * mesh[1] * (mesh[2] / 2 + 1));
    if (!Zh) return ISDF_ERR_HIP;
  }
  for (int r = row0; r < row0 + nrows; r += batch) {
    const int nb = std::min(batch, row0 + nrows - r);
    int rc = ISDF_OK;
    if (own) {
      rc = conv_rows_q_own(h, d_theta + (int64_t)r * ldt, Vre, Vim, nb, mesh, d_coulG, Zh, Z);
      if (rc) return rc;
    } else {
    hipfftHandle plan;
    rc = get_z2z_plan(h, mesh, nb, &plan);
    if (rc) return rc;
    const int64_t total = (int64_t)nb * G;
    const unsigned nblocks = (unsigned)std::min<int64_t>(cdiv(total, 256), (int64_t)h->num_cu * 16);
    {
      ProfScope ps(h, "coulomb_conv_z2z[byte]", 64.0 * (double)total, 5);
      hipLaunchKernelGGL(pack_real_to_complex_kernel, dim3(nblocks), dim3(256), 0, h->stream,
                         d_theta + (int64_t)r * ldt, Z, total);
      FFT_TRY(h, hipfftExecZ2Z(plan, (hipfftDoubleComplex*)Z, (hipfftDoubleComplex*)Z, HIPFFT_FORWARD));
      hipLaunchKernelGGL(mul_full_kernel, dim3(nblocks), dim3(256), 0, h->stream, Z, d_coulG, G, total, 1.0 / (double)G);
      FFT_TRY(h, hipfftExecZ2Z(plan, (hipfftDoubleComplex*)Z, (hipfftDoubleComplex*)Z, HIPFFT_BACKWARD));
      hipLaunchKernelGGL(unpack_complex_kernel, dim3(nblocks), dim3(256), 0, h->stream, Z, Vre, Vim, total);
      KERNEL_CHECK(h);
    }
    }
    const int c0 = upper_only ? r : 0;
    rc = gemm_nt_f64(h, nb, P - c0, G, weight, Vre, G, d_theta + (int64_t)c0 * ldt, ldt, 0.0,
                     d_Wre + (int64_t)r * ldw + c0, ldw);
    if (rc) return rc;
    rc = gemm_nt_f64(h, nb, P - c0, G, weight, Vim, G, d_theta + (int64_t)c0 * ldt, ldt, 0.0,
                     d_Wim + (int64_t)r * ldw + c0, ldw);
    if (rc) return rc;
  }
  return ISDF_OK;
}

extern "C" int isdf_coulomb_rows_q(isdf_handle h, const double* d_rows, int nrows, int64_t ld, const int32_t mesh[3],
                                   const double* d_coulG, double* d_re, double* d_im) {
  // (d_re + i d_im)[r] = ifft(coulG(q) fft(rows[r])): the k-point convolution of real rows with a full real kernel table, rows of
  // G contiguous (ld == G) - V^q = conv_q(Theta) for the robust K at k-points; the convolution step of isdf_coulomb_Wq on its own
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_rows && mesh && d_coulG && d_re && d_im && nrows > 0);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  ARG_CHECK(h, ld == G);
  double2* Z = (double2*)isdf_ws(h, "coul_Zfull", sizeof(double2) * (size_t)nrows * G);
  if (!Z) return ISDF_ERR_HIP;
  if (conv_rows_q_own_supported(h, mesh, nrows)) {
    double2* Zh = (double2*)isdf_ws(h, "coul_Z", sizeof(double2) * (size_t)nrows * mesh[0] * mesh[1] * (mesh[2] / 2 + 1));
    if (!Zh) return ISDF_ERR_HIP;
    return conv_rows_q_own(h, d_rows, d_re, d_im, nrows, mesh, d_coulG, Zh, Z);
  }
  hipfftHandle plan;
  int rc = get_z2z_plan(h, mesh, nrows, &plan);
  if (rc) return rc;
  const int64_t total = (int64_t)nrows * G;
  const unsigned nblocks = (unsigned)std::min<int64_t>(cdiv(total, 256), (int64_t)h->num_cu * 16);
  ProfScope ps(h, "coulomb_conv_z2z[byte]", 64.0 * (double)total, 5);
  hipLaunchKernelGGL(pack_real_to_complex_kernel, dim3(nblocks), dim3(256), 0, h->stream, d_rows, Z, total);
  FFT_TRY(h, hipfftExecZ2Z(plan, (hipfftDoubleComplex*)Z, (hipfftDoubleComplex*)Z, HIPFFT_FORWARD));
  hipLaunchKernelGGL(mul_full_kernel, dim3(nblocks), dim3(256), 0, h->stream, Z, d_coulG, G, total, 1.0 / (double)G);
  FFT_TRY(h, hipfftExecZ2Z(plan, (hipfftDoubleComplex*)Z, (hipfftDoubleComplex*)Z, HIPFFT_BACKWARD));
  hipLaunchKernelGGL(unpack_complex_kernel, dim3(nblocks), dim3(256), 0, h->stream, Z, d_re, d_im, total);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_nyquist_spectra(isdf_handle h, const double* d_rows, int nrows, int64_t ld, const int32_t mesh[3], int axis,
                                    double* d_re, double* d_im) {
  // (d_re + i d_im)[r][ka][kb] = the 3-D DFT of row r on the NYQUIST plane of `axis` (index mesh[axis]/2, mesh[axis] even): the
  // alternating-sign sum along that axis followed by a 2-D transform over the other two (their natural order, C layout).  Used for
  // the even-mesh correction of the +-q pairing of the k-point build.
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_rows && mesh && d_re && d_im && nrows > 0 && nrows <= 65535 && axis >= 0 && axis <= 2 && mesh[axis] % 2 == 0);
  const int n0 = mesh[0], n1 = mesh[1], n2 = mesh[2];
  ARG_CHECK(h, ld >= (int64_t)n0 * n1 * n2);
  const int na = axis == 0 ? n1 : n0, nb = axis == 2 ? n1 : n2;
  const int64_t np = (int64_t)na * nb;
  double2* Z = (double2*)isdf_ws(h, "nyq_Z", sizeof(double2) * (size_t)nrows * np);
  if (!Z) return ISDF_ERR_HIP;
  std::vector<int> key = {na, nb, nrows, -2, 0};
  auto it = h->plans.find(key);
  hipfftHandle plan;
  if (it != h->plans.end()) plan = it->second.fwd;
  else {
    FftPlan p;
    int dims[2] = {na, nb};
    FFT_TRY(h, hipfftPlanMany(&p.fwd, 2, dims, nullptr, 1, (int)np, nullptr, 1, (int)np, HIPFFT_Z2Z, nrows));
    FFT_TRY(h, hipfftSetStream(p.fwd, h->stream));
    p.bwd = 0;
    h->plans.emplace(key, p);
    plan = p.fwd;
  }
  hipLaunchKernelGGL(nyquist_reduce_kernel, dim3((unsigned)cdiv(np, 256), (unsigned)nrows), dim3(256), 0, h->stream, d_rows, ld, n0, n1,
                     n2, axis, Z);
  FFT_TRY(h, hipfftExecZ2Z(plan, (hipfftDoubleComplex*)Z, (hipfftDoubleComplex*)Z, HIPFFT_FORWARD));
  const int64_t total = (int64_t)nrows * np;
  hipLaunchKernelGGL(unpack_complex_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), (int64_t)h->num_cu * 16)), dim3(256), 0,
                     h->stream, Z, d_re, d_im, total);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_zhadamard_planes(isdf_handle h, double* d_Ar, double* d_Ai, int64_t lda, const double* d_Br, const double* d_Bi,
                                     int64_t ldb, int rows, int64_t cols) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_Ar && d_Ai && d_Br && d_Bi && rows > 0 && rows <= 65535 && cols > 0 && lda >= cols && ldb >= cols);
  hipLaunchKernelGGL(zhadamard_planes_kernel, dim3((unsigned)cdiv(cols, 256), (unsigned)rows), dim3(256), 0, h->stream, d_Ar, d_Ai,
                     lda, d_Br, d_Bi, ldb, cols);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_symmetrize_hermitian(isdf_handle h, double* d_Wre, double* d_Wim, int P, int64_t ldw) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_Wre && d_Wim && P > 0 && ldw >= P);
  const unsigned nt = (unsigned)cdiv(P, 32);
  hipLaunchKernelGGL(mirror_upper_sign_kernel, dim3(nt, nt), dim3(32, 8), 0, h->stream, d_Wre, P, ldw, 1.0);
  hipLaunchKernelGGL(mirror_upper_sign_kernel, dim3(nt, nt), dim3(32, 8), 0, h->stream, d_Wim, P, ldw, -1.0);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_finish_Wq(isdf_handle h, const double* d_Wre, const double* d_Wim, int P, int64_t ldw,
                              const double* d_phase, double* d_Wc) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_Wre && d_Wim && d_phase && d_Wc && P > 0 && P <= 65535 && ldw >= P);
  hipLaunchKernelGGL(finish_Wq_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)P), dim3(256), 0, h->stream, d_Wre,
                     d_Wim, P, ldw, (const double2*)d_phase, (double2*)d_Wc);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_get_k_pair(isdf_handle h, const double* d_A1, const double* d_A2, const double* d_D2,
                               const double* d_Wq, int P, int nao, double scale, double* d_vk) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_A1 && d_A2 && d_D2 && d_Wq && d_vk && P > 0 && nao > 0);
  double2* T = (double2*)isdf_ws(h, "kk_T", sizeof(double2) * (size_t)P * nao);
  double2* X = (double2*)isdf_ws(h, "kk_X", sizeof(double2) * (size_t)P * P);
  double2* Y = (double2*)isdf_ws(h, "kk_Y", sizeof(double2) * (size_t)P * nao);
  if (!T || !X || !Y) return ISDF_ERR_HIP;
  const double2 one = make_double2(1.0, 0.0), zero = make_double2(0.0, 0.0);
  const double2* A1 = (const double2*)d_A1;
  const double2* A2 = (const double2*)d_A2;
  int rc = zgemm_rm(h, 'N', 'N', P, nao, nao, one, A2, nao, (const double2*)d_D2, nao, zero, T, nao);
  if (rc) return rc;
  rc = zgemm_rm(h, 'N', 'C', P, P, nao, one, T, nao, A2, nao, zero, X, P);
  if (rc) return rc;
  const int64_t n = (int64_t)P * P;
  hipLaunchKernelGGL(zhadamard_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), (int64_t)h->num_cu * 16)), dim3(256),
                     0, h->stream, X, (const double2*)d_Wq, n);
  KERNEL_CHECK(h);
  rc = zgemm_rm(h, 'N', 'N', P, nao, P, one, X, P, A1, nao, zero, Y, nao);
  if (rc) return rc;
  return zgemm_rm(h, 'C', 'N', nao, nao, P, make_double2(scale, 0.0), A1, nao, Y, nao, one, (double2*)d_vk, nao);
}

extern "C" int isdf_get_k_exact_kpt(isdf_handle h, const double* d_u1r, const double* d_u1i, int nao, int64_t ld1,
                                    const double* d_m2r, const double* d_m2i, int nocc, int64_t ld2, const int32_t mesh[3],
                                    const double* d_coulG, double weight, int i0, int ni, int max_rows, double* d_vk_re,
                                    double* d_vk_im) {
  // The reference's exact k-point exchange for ONE (k1, k2) pair (pyscf/pbc/df/fft_jk.py:250-292), in periodic parts:
  //   vk[p, k'] += weight sum_g { sum_j conv_q[conj(u1_p) m2_j](g) conj(m2_j(g)) } u1_k'(g),   p in [i0, i0 + ni)
  // u1 (nao, G): periodic parts of the Bloch AOs at k1 (two real planes), m2 (nocc, G): those of the occupied orbitals at k2
  // (scaled with sqrt(occ)), d_coulG: the kernel table of q = k2 - k1 (isdf_coulG_q), weight = vol / G / nk.  One complex FFT pair
  // per (AO, occupied orbital) pair - the verification path for the k-point ISDF exchange.
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_u1r && d_u1i && d_m2r && d_m2i && mesh && d_coulG && d_vk_re && d_vk_im && nao > 0 && nocc > 0);
  ARG_CHECK(h, i0 >= 0 && ni >= 0 && i0 + ni <= nao && max_rows >= nocc && nocc <= 65535);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  ARG_CHECK(h, ld1 >= G && ld2 >= G);
  if (ni == 0) return ISDF_OK;
  const int bi = std::max(1, std::min(ni, max_rows / nocc));
  ARG_CHECK(h, (int64_t)bi * nocc <= 65535);
  double2* Z = (double2*)isdf_ws(h, "kxk_Z", sizeof(double2) * (size_t)bi * nocc * G);
  double* Tr = (double*)isdf_ws(h, "kxk_T", sizeof(double) * (size_t)2 * bi * G);
  if (!Z || !Tr) return ISDF_ERR_HIP;
  double* Ti = Tr + (size_t)bi * G;
  for (int r = i0; r < i0 + ni; r += bi) {
    const int nb = std::min(bi, i0 + ni - r);
    const int rows = nb * nocc;
    hipfftHandle plan;
    int rc = get_z2z_plan(h, mesh, rows, &plan);
    if (rc) return rc;
    const int64_t total = (int64_t)rows * G;
    const unsigned nblocks = (unsigned)std::min<int64_t>(cdiv(total, 256), (int64_t)h->num_cu * 16);
    {
      ProfScope ps(h, "exact_k_kpt_pairs[byte]", 96.0 * (double)total, 5);
      hipLaunchKernelGGL(pair_rows_cplx_kernel, dim3((unsigned)cdiv(G, 256), (unsigned)rows), dim3(256), 0, h->stream,
                         d_u1r + (int64_t)r * ld1, d_u1i + (int64_t)r * ld1, ld1, d_m2r, d_m2i, ld2, nocc, G, Z);
      FFT_TRY(h, hipfftExecZ2Z(plan, (hipfftDoubleComplex*)Z, (hipfftDoubleComplex*)Z, HIPFFT_FORWARD));
      hipLaunchKernelGGL(mul_full_kernel, dim3(nblocks), dim3(256), 0, h->stream, Z, d_coulG, G, total, 1.0 / (double)G);
      FFT_TRY(h, hipfftExecZ2Z(plan, (hipfftDoubleComplex*)Z, (hipfftDoubleComplex*)Z, HIPFFT_BACKWARD));
      hipLaunchKernelGGL(pair_reduce_cplx_kernel, dim3((unsigned)cdiv(G, 256), (unsigned)nb), dim3(256), 0, h->stream, Z, d_m2r,
                         d_m2i, ld2, nocc, G, Tr, Ti);
      KERNEL_CHECK(h);
    }
    // vk[r:r+nb] += weight T u1^T (complex, no conjugate):  Re = Tr u1r^T - Ti u1i^T,  Im = Tr u1i^T + Ti u1r^T
    double* kr = d_vk_re + (int64_t)(r - i0) * nao;
    double* ki = d_vk_im + (int64_t)(r - i0) * nao;
    rc = gemm_nt_f64(h, nb, nao, G, weight, Tr, G, d_u1r, ld1, 1.0, kr, nao);
    if (rc) return rc;
    rc = gemm_nt_f64(h, nb, nao, G, -weight, Ti, G, d_u1i, ld1, 1.0, kr, nao);
    if (rc) return rc;
    rc = gemm_nt_f64(h, nb, nao, G, weight, Tr, G, d_u1i, ld1, 1.0, ki, nao);
    if (rc) return rc;
    rc = gemm_nt_f64(h, nb, nao, G, weight, Ti, G, d_u1r, ld1, 1.0, ki, nao);
    if (rc) return rc;
  }
  return ISDF_OK;
}

extern "C" int isdf_rho_k(isdf_handle h, const double* d_ur, const double* d_ui, int nao, int64_t ng, int64_t ld,
                          const double* d_DTr, const double* d_DTi, double scale, double* d_rho) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_ur && d_ui && d_DTr && d_DTi && d_rho && nao > 0 && ng > 0 && ld >= ng);
  const int64_t CH = 32768;
  double* Tr = (double*)isdf_ws(h, "jk_T", sizeof(double) * (size_t)2 * nao * CH);
  if (!Tr) return ISDF_ERR_HIP;
  double* Ti = Tr + (size_t)nao * CH;
  for (int64_t g0 = 0; g0 < ng; g0 += CH) {
    const int64_t nc = std::min(CH, ng - g0);
    // T = D^T u (complex): Tr = DTr ur - DTi ui, Ti = DTr ui + DTi ur
    int rc = gemm_rm(h, 'N', 'N', nao, nc, nao, 1.0, d_DTr, nao, d_ur + g0, ld, 0.0, Tr, CH);
    if (rc) return rc;
    rc = gemm_rm(h, 'N', 'N', nao, nc, nao, -1.0, d_DTi, nao, d_ui + g0, ld, 1.0, Tr, CH);
    if (rc) return rc;
    rc = gemm_rm(h, 'N', 'N', nao, nc, nao, 1.0, d_DTr, nao, d_ui + g0, ld, 0.0, Ti, CH);
    if (rc) return rc;
    rc = gemm_rm(h, 'N', 'N', nao, nc, nao, 1.0, d_DTi, nao, d_ur + g0, ld, 1.0, Ti, CH);
    if (rc) return rc;
    hipLaunchKernelGGL(rho_k_reduce_kernel, dim3((unsigned)cdiv(nc, 256)), dim3(256), 0, h->stream, Tr, Ti, CH,
                       d_ur + g0, d_ui + g0, ld, nao, nc, scale, d_rho + g0);
    KERNEL_CHECK(h);
  }
  return ISDF_OK;
}

extern "C" int isdf_vj_k(isdf_handle h, const double* d_ur, const double* d_ui, int nao, int64_t ng, int64_t ld,
                         const double* d_vR, double* d_vj_re, double* d_vj_im) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_ur && d_ui && d_vR && d_vj_re && d_vj_im && nao > 0 && ng > 0 && ld >= ng);
  // vj_ij = sum_g conj(u_i) v u_j :  Re = ur (v ur)^T + ui (v ui)^T,  Im = ur (v ui)^T - ui (v ur)^T
  int rc = gemm_nt_f64_scaled(h, nao, nao, ng, 1.0, d_ur, ld, d_ur, ld, d_vR, 0.0, d_vj_re, nao);
  if (rc) return rc;
  rc = gemm_nt_f64_scaled(h, nao, nao, ng, 1.0, d_ui, ld, d_ui, ld, d_vR, 1.0, d_vj_re, nao);
  if (rc) return rc;
  rc = gemm_nt_f64_scaled(h, nao, nao, ng, 1.0, d_ur, ld, d_ui, ld, d_vR, 0.0, d_vj_im, nao);
  if (rc) return rc;
  return gemm_nt_f64_scaled(h, nao, nao, ng, -1.0, d_ui, ld, d_ur, ld, d_vR, 1.0, d_vj_im, nao);
}

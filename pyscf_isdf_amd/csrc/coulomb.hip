// S4+S5: FFT-based Coulomb convolution of the interpolation vectors and W = (vol/G) V Theta^T.
//
// Per batch of rows: batched real-to-complex 3-D FFT (hipFFT D2Z) -> multiply the half spectrum by
// coulG/G (fused scaling: the reference's ifft carries the 1/G, pyscf/pbc/tools/pbc.py:182-211) ->
// complex-to-real inverse (Z2D) -> contraction with Theta on the matrix cores (gemm_nt_f64).
// coulG = 4 pi / |G|^2, G = 0 -> 0 (pbc.py:352-356); G vectors in fftfreq order (cell.py:552-587).
// The table is symmetrised (see coulG_half_kernel) so that the real-to-complex / complex-to-real pair
// reproduces the reference's complex transform + .real exactly, also on even, non-orthogonal meshes.
#include "common.h"

namespace {

// |G|^2 of mesh index (ix,iy,iz) with numpy.fft.fftfreq(n, 1/n) frequencies: 0..(n-1)/2, -(n/2)..-1
struct Recip { double b[9]; };
__device__ inline double g2_of(int ix, int iy, int iz, int n0, int n1, int n2, const Recip& r) {
  const double fx = (ix < (n0 + 1) / 2) ? ix : ix - n0;
  const double fy = (iy < (n1 + 1) / 2) ? iy : iy - n1;
  const double fz = (iz < (n2 + 1) / 2) ? iz : iz - n2;
  const double gx = fx * r.b[0] + fy * r.b[3] + fz * r.b[6];
  const double gy = fx * r.b[1] + fy * r.b[4] + fz * r.b[7];
  const double gz = fx * r.b[2] + fy * r.b[5] + fz * r.b[8];
  return gx * gx + gy * gy + gz * gz;
}

// Half-spectrum table (n0, n1, n2/2+1) of the kernel the reference effectively applies:
// Re[ifft(c * fft(rho))] for real rho equals ifft(c_sym * fft(rho)) with
// c_sym(G) = (c(idx) + c(mirror idx)) / 2.  On even meshes the Nyquist index keeps frequency -n/2
// in both idx and its mirror, so c is not inversion symmetric there for non-orthogonal lattices;
// symmetrising makes the real-to-complex / complex-to-real pair reproduce the reference exactly.
// omega != 0: range separation as pyscf/pbc/tools/pbc.py:408-418 applies it to the kernel, 4 pi/G^2 * exp(-G^2/(4 omega^2))
// for omega > 0 (long range, erf(omega r)/r), 4 pi/G^2 * (1 - exp(-G^2/(4 omega^2))) for omega < 0 (short range); G = 0 stays 0.
__device__ inline double range_factor(double g2, double omega) {
  if (omega == 0.0) return 1.0;
  const double e = exp(-0.25 * g2 / (omega * omega));
  return omega > 0.0 ? e : 1.0 - e;
}
// 4 pi / G^2 times the range-separation factor, or - rc > 0, exxdiv='vcut_sph' (pbc.py:312-317, PRB 77 193110) - times
// 1 - cos(|G| rc), the transform of 1/r cut at r = rc; its G -> 0 limit is 2 pi rc^2
__device__ inline double kernel_value(double g2, double omega, double rc) {
  const double fourpi = 4.0 * 3.14159265358979323846;
  if (rc > 0.0) return g2 == 0.0 ? 0.5 * fourpi * rc * rc : fourpi / g2 * (1.0 - cos(sqrt(g2) * rc));
  return g2 == 0.0 ? 0.0 : fourpi / g2 * range_factor(g2, omega);
}

__global__ void coulG_half_kernel(double* __restrict__ out, int n0, int n1, int n2, Recip r,
                                  double scale, double omega, double rc, double g2cut) {
  const int n2h = n2 / 2 + 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tot = (int64_t)n0 * n1 * n2h;
  if (idx >= tot) return;
  const int iz = (int)(idx % n2h);
  const int iy = (int)((idx / n2h) % n1);
  const int ix = (int)(idx / ((int64_t)n2h * n1));
  if (ix == 0 && iy == 0 && iz == 0) { out[idx] = scale * kernel_value(0.0, omega, rc); return; }
  const double g1 = g2_of(ix, iy, iz, n0, n1, n2, r);
  const double g2 = g2_of((n0 - ix) % n0, (n1 - iy) % n1, (n2 - iz) % n2, n0, n1, n2, r);
  // g2cut > 0 (option "coul_sphere"): entries beyond the sphere |G|^2 <= g2cut are dropped
  if (g2cut > 0.0 && (g1 > g2cut || g2 > g2cut)) { out[idx] = 0.0; return; }
  out[idx] = scale * 0.5 * (kernel_value(g1, omega, rc) + kernel_value(g2, omega, rc));
}

// Full-spectrum kernel table for a difference vector q (k-point exchange, pyscf/pbc/tools/pbc.py:230-420 with exxdiv=None),
// worked out in INDEX space: with q = sum_i qf_i b_i and the mesh frequency f_i, the component of q + G along b_i in units of
// the mesh edge (n_i//2 + 1/2) b_i is x_i = (f_i + qf_i) / (n_i//2 + 1/2).  Components with |x_i| >= 1 (to 9 decimals, the
// reference's rounding) lie beyond the edge and are wrapped back by 2 (n_i//2) + 1 frequencies (pbc.py:272-302); a component
// exactly ON the edge (|x_i| = 1) is ambiguous and its table entry is zeroed (pbc.py:400-401).
// exxdiv='vcut_ws': short-range part 4 pi/g^2 (1 - exp(-g^2/4 alpha^2)) (g = 0: pi/alpha^2) plus the precomputed table of
// the long-range part truncated to the Wigner-Seitz cell, looked up at the integer coordinates of k + G on the nk-fold
// cell's reciprocal lattice (pbc.py:318-346) for vectors inside the table's range
__device__ inline double ws_kernel_value(double gx, double gy, double gz, double g2, const WsKernel& ws) {
  const double pi = 3.14159265358979323846;
  double v = g2 == 0.0 ? pi / (ws.alpha * ws.alpha) : 4.0 * pi / g2 * (1.0 - exp(-g2 / (4.0 * ws.alpha * ws.alpha)));
  if (fabs(gx) <= ws.maxq[0] && fabs(gy) <= ws.maxq[1] && fabs(gz) <= ws.maxq[2]) {
    int id[3];
    for (int d = 0; d < 3; ++d) {
      const double t = (gx * ws.ak[3 * d] + gy * ws.ak[3 * d + 1] + gz * ws.ak[3 * d + 2]) / (2.0 * pi);
      const int gi = (int)(rint(t * 1e6) / 1e6);                 // round(6 decimals) then truncation, as the reference
      id[d] = ((gi % ws.mesh[d]) + ws.mesh[d]) % ws.mesh[d];
    }
    v += ws.vq[((int64_t)id[0] * ws.mesh[1] + id[1]) * ws.mesh[2] + id[2]];
  }
  return v;
}

__global__ void half_from_full_kernel(const double* __restrict__ full, double* __restrict__ out, int n0, int n1, int n2,
                                      double scale) {
  // symmetrised half-spectrum table from a full table: c_sym = (c(idx) + c(mirror idx)) / 2  (see coulG_half_kernel)
  const int n2h = n2 / 2 + 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n0 * n1 * n2h) return;
  const int iz = (int)(idx % n2h);
  const int iy = (int)((idx / n2h) % n1);
  const int ix = (int)(idx / ((int64_t)n2h * n1));
  const int64_t a = ((int64_t)ix * n1 + iy) * n2 + iz;
  const int64_t b = ((int64_t)((n0 - ix) % n0) * n1 + (n1 - iy) % n1) * n2 + (n2 - iz) % n2;
  out[idx] = scale * 0.5 * (full[a] + full[b]);
}

__global__ void coulG_q_kernel(double* __restrict__ out, int n0, int n1, int n2, Recip r, double qf0, double qf1,
                               double qf2, int wrap, double omega, double rc, WsKernel ws) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tot = (int64_t)n0 * n1 * n2;
  if (idx >= tot) return;
  const int iz = (int)(idx % n2);
  const int iy = (int)((idx / n2) % n1);
  const int ix = (int)(idx / ((int64_t)n2 * n1));
  const int n[3] = {n0, n1, n2};
  const int id[3] = {ix, iy, iz};
  const double qf[3] = {qf0, qf1, qf2};
  double c[3];
  bool edge = false;
  for (int d = 0; d < 3; ++d) {
    double f = (id[d] < (n[d] + 1) / 2) ? id[d] : id[d] - n[d];
    if (wrap) {
      const int half = n[d] / 2;
      const double x9 = rint((f + qf[d]) / (half + 0.5) * 1e9);   // numpy.round(x, 9) * 1e9
      if (x9 >= 1e9) f -= 2 * half + 1;
      else if (x9 <= -1e9) f += 2 * half + 1;
      edge = edge || x9 == 1e9 || x9 == -1e9;
    }
    c[d] = f + qf[d];
  }
  const double gx = c[0] * r.b[0] + c[1] * r.b[3] + c[2] * r.b[6];
  const double gy = c[0] * r.b[1] + c[1] * r.b[4] + c[2] * r.b[7];
  const double gz = c[0] * r.b[2] + c[1] * r.b[5] + c[2] * r.b[8];
  const double g2 = gx * gx + gy * gy + gz * gz;
  out[idx] = edge ? 0.0 : (ws.alpha > 0.0 ? ws_kernel_value(gx, gy, gz, g2, ws) : kernel_value(g2, omega, rc));
}

__global__ void mul_half_kernel(double2* __restrict__ z, const double* __restrict__ cg, int64_t gc,
                                int64_t total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const double c = cg[i % gc];
    double2 v = z[i];
    v.x *= c;
    v.y *= c;
    z[i] = v;
  }
}

}  // namespace

// Build (or fetch) the scaled half-spectrum Coulomb table for this mesh/lattice.
int get_coulG_half(isdf_handle h, const int32_t mesh[3], const double a[9], double extra_scale, double** out) {
  const int64_t gc = (int64_t)mesh[0] * mesh[1] * (mesh[2] / 2 + 1);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  double* cg = (double*)isdf_ws(h, "coulG_half", sizeof(double) * gc);
  if (!cg) return ISDF_ERR_HIP;
  // b = 2 pi inv(a^T): rows b_i (pyscf/pbc/gto/cell.py:1571-1591)
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
                     a[2] * (a[3] * a[7] - a[4] * a[6]);
  if (det == 0.0) return isdf_fail(h, ISDF_ERR_ARG, "singular lattice");
  const double tp = 2.0 * 3.14159265358979323846 / det;
  // b_0 = 2pi (a1 x a2)/det, b_1 = 2pi (a2 x a0)/det, b_2 = 2pi (a0 x a1)/det
  const double b[9] = {tp * (a[4] * a[8] - a[5] * a[7]), tp * (a[5] * a[6] - a[3] * a[8]), tp * (a[3] * a[7] - a[4] * a[6]),
                       tp * (a[7] * a[2] - a[8] * a[1]), tp * (a[8] * a[0] - a[6] * a[2]), tp * (a[6] * a[1] - a[7] * a[0]),
                       tp * (a[1] * a[5] - a[2] * a[4]), tp * (a[2] * a[3] - a[0] * a[5]), tp * (a[0] * a[4] - a[1] * a[3])};
  Recip rr;
  for (int i = 0; i < 9; ++i) rr.b[i] = b[i];
  if (h->wsk.alpha > 0.0) {
    // tabulated kernel (exxdiv='vcut_ws'): full table at q = 0, then the symmetrised half spectrum
    double* full = (double*)isdf_ws(h, "coulG_full", sizeof(double) * G);
    if (!full) return ISDF_ERR_HIP;
    hipLaunchKernelGGL(coulG_q_kernel, dim3((unsigned)cdiv(G, 256)), dim3(256), 0, h->stream, full, mesh[0], mesh[1], mesh[2],
                       rr, 0.0, 0.0, 0.0, 0, 0.0, 0.0, h->wsk);
    hipLaunchKernelGGL(half_from_full_kernel, dim3((unsigned)cdiv(gc, 256)), dim3(256), 0, h->stream, full, cg, mesh[0], mesh[1],
                       mesh[2], extra_scale / (double)G);
    KERNEL_CHECK(h);
    *out = cg;
    return ISDF_OK;
  }
  double g2cut = 0.0;
  if (h->coul_sphere > 0) {
    // radius of the sphere inscribed in the reciprocal FFT box (faces kappa_i = +-(n_i - 1) / 2, at distance 2 pi kappa_i / |a_i|),
    // times coul_sphere percent
    double rmin = 1e300;
    for (int i = 0; i < 3; ++i) {
      const double an = sqrt(a[3 * i] * a[3 * i] + a[3 * i + 1] * a[3 * i + 1] + a[3 * i + 2] * a[3 * i + 2]);
      rmin = std::min(rmin, 2.0 * 3.14159265358979323846 * (double)((mesh[i] - 1) / 2) / an);
    }
    rmin *= 0.01 * (double)h->coul_sphere;
    g2cut = rmin * rmin * (1.0 + 1e-12);
  }
  hipLaunchKernelGGL(coulG_half_kernel, dim3((unsigned)cdiv(gc, 256)), dim3(256), 0, h->stream, cg, mesh[0],
                     mesh[1], mesh[2], rr, extra_scale / (double)G, h->coul_omega, h->coul_rc, g2cut);
  KERNEL_CHECK(h);
  *out = cg;
  return ISDF_OK;
}

static int reciprocal_rows(isdf_handle h, const double a[9], Recip* rr) {
  // b = 2 pi inv(a^T): rows b_i (pyscf/pbc/gto/cell.py:1571-1591)
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
                     a[2] * (a[3] * a[7] - a[4] * a[6]);
  if (det == 0.0) return isdf_fail(h, ISDF_ERR_ARG, "singular lattice");
  const double tp = 2.0 * 3.14159265358979323846 / det;
  const double b[9] = {tp * (a[4] * a[8] - a[5] * a[7]), tp * (a[5] * a[6] - a[3] * a[8]), tp * (a[3] * a[7] - a[4] * a[6]),
                       tp * (a[7] * a[2] - a[8] * a[1]), tp * (a[8] * a[0] - a[6] * a[2]), tp * (a[6] * a[1] - a[7] * a[0]),
                       tp * (a[1] * a[5] - a[2] * a[4]), tp * (a[2] * a[3] - a[0] * a[5]), tp * (a[0] * a[4] - a[1] * a[3])};
  for (int i = 0; i < 9; ++i) rr->b[i] = b[i];
  return ISDF_OK;
}

extern "C" int isdf_coulG_q(isdf_handle h, const int32_t mesh[3], const double a[9], const double q[3], int wrap_around,
                            double omega, double* d_out) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, mesh && a && q && d_out && mesh[0] > 0 && mesh[1] > 0 && mesh[2] > 0);
  Recip rr;
  int rc = reciprocal_rows(h, a, &rr);
  if (rc) return rc;
  // q in units of the reciprocal vectors: q = qf b  ->  qf = q a^T / 2 pi
  double qf[3];
  for (int i = 0; i < 3; ++i)
    qf[i] = (q[0] * a[3 * i] + q[1] * a[3 * i + 1] + q[2] * a[3 * i + 2]) / (2.0 * 3.14159265358979323846);
  const bool nonzero = fabs(q[0]) + fabs(q[1]) + fabs(q[2]) > 1e-9;     // the reference wraps only for q != 0 (pbc.py:272)
  const int wrap = (wrap_around && nonzero) ? 1 : 0;
  if (wrap)
    for (int i = 0; i < 3; ++i)
      if (fabs(rint(qf[i] / (mesh[i] / 2 + 0.5) * 1e9)) >= 1e9)
        return isdf_fail(h, ISDF_ERR_ARG, "isdf_coulG_q: q lies outside the first FFT box (pyscf/pbc/tools/pbc.py:281)");
  if (!nonzero) qf[0] = qf[1] = qf[2] = 0.0;
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  hipLaunchKernelGGL(coulG_q_kernel, dim3((unsigned)cdiv(G, 256)), dim3(256), 0, h->stream, d_out, mesh[0], mesh[1],
                     mesh[2], rr, qf[0], qf[1], qf[2], wrap, omega, h->coul_rc, h->wsk);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

// In-place Coulomb convolution of nb real fields (rows of length G, contiguous): the FFT buffer
// `zbuf` must hold nb * gc complex numbers.
static int convolve_rows(isdf_handle h, const double* d_in, double* d_out, int nb, const int32_t mesh[3],
                         const double* cg, double2* zbuf) {
  const int64_t gc = (int64_t)mesh[0] * mesh[1] * (mesh[2] / 2 + 1);
  if (h->own_fft && conv_rows_own_supported(mesh, nb)) return conv_rows_own(h, d_in, d_out, nb, mesh, cg, zbuf);
  FftPlan* plan = nullptr;
  int rc = isdf_get_plan(h, mesh, nb, &plan);
  if (rc) return rc;
  const int64_t Gfull = (int64_t)mesh[0] * mesh[1] * mesh[2];
  ProfScope ps(h, "coulomb_conv_d2z_mul_z2d[byte]", 32.0 * (double)Gfull * nb, 3);
  FFT_TRY(h, hipfftExecD2Z(plan->fwd, (hipfftDoubleReal*)d_in, (hipfftDoubleComplex*)zbuf));
  const int64_t total = gc * nb;
  const unsigned nblocks = (unsigned)std::min<int64_t>(cdiv(total, 256), (int64_t)h->num_cu * 16);
  hipLaunchKernelGGL(mul_half_kernel, dim3(nblocks), dim3(256), 0, h->stream, zbuf, cg, gc, total);
  KERNEL_CHECK(h);
  FFT_TRY(h, hipfftExecZ2D(plan->bwd, (hipfftDoubleComplex*)zbuf, (hipfftDoubleReal*)d_out));
  return ISDF_OK;
}

namespace {
// W[q][p] = W[p][q] for q > p (the Coulomb operator is symmetric; only the block-upper part is computed)
__global__ void mirror_upper_kernel(double* __restrict__ W, int P, int64_t ldw) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x, by = blockIdx.y;     // tile (by, bx) of the upper triangle, bx >= by
  if (bx < by) return;
  const int tx = threadIdx.x, ty = threadIdx.y;   // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = by * 32 + i, cc = bx * 32 + tx;
    tile[i][tx] = (r < P && cc < P) ? W[(int64_t)r * ldw + cc] : 0.0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int r = bx * 32 + i, cc = by * 32 + tx;   // transposed position
    if (r < P && cc < P && r > cc) W[(int64_t)r * ldw + cc] = tile[tx][i];
  }
}
// W <- (W + sign W^T) / 2, tile pairs (by, bx) / (bx, by) with bx >= by
__global__ void mean_symmetric_kernel(double* __restrict__ W, int P, int64_t ldw, double sign) {
  __shared__ double up[32][33], lo[32][33];
  const int bx = blockIdx.x, by = blockIdx.y;
  if (bx < by) return;
  const int tx = threadIdx.x, ty = threadIdx.y;   // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = by * 32 + i, cc = bx * 32 + tx;
    up[i][tx] = (r < P && cc < P) ? W[(int64_t)r * ldw + cc] : 0.0;
    const int r2 = bx * 32 + i, c2 = by * 32 + tx;
    lo[i][tx] = (r2 < P && c2 < P) ? W[(int64_t)r2 * ldw + c2] : 0.0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int r = by * 32 + i, cc = bx * 32 + tx;
    if (r < P && cc < P) W[(int64_t)r * ldw + cc] = 0.5 * (up[i][tx] + sign * lo[tx][i]);
    const int r2 = bx * 32 + i, c2 = by * 32 + tx;
    if (bx != by && r2 < P && c2 < P) W[(int64_t)r2 * ldw + c2] = 0.5 * (lo[i][tx] + sign * up[tx][i]);
  }
}
}  // namespace

extern "C" int isdf_symmetrize_mean(isdf_handle h, double* d_W, int P, int64_t ldw, int antisymmetric) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_W && P > 0 && ldw >= P);
  const unsigned nt = (unsigned)cdiv(P, 32);
  hipLaunchKernelGGL(mean_symmetric_kernel, dim3(nt, nt), dim3(32, 8), 0, h->stream, d_W, P, ldw,
                     antisymmetric ? -1.0 : 1.0);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_symmetrize_upper(isdf_handle h, double* d_W, int P, int64_t ldw) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_W && P > 0 && ldw >= P);
  const unsigned nt = (unsigned)cdiv(P, 32);
  hipLaunchKernelGGL(mirror_upper_kernel, dim3(nt, nt), dim3(32, 8), 0, h->stream, d_W, P, ldw);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_coulomb_W(isdf_handle h, const double* d_theta, int P, int64_t ldt,
                              const int32_t mesh[3], const double a[9], int row0, int nrows,
                              int batch, int upper_only, double* d_W, int64_t ldw) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_theta && mesh && a && d_W && P > 0 && batch > 0 && ldw >= P);
  ARG_CHECK(h, row0 >= 0 && nrows >= 0 && row0 + nrows <= P);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  const int64_t gc = (int64_t)mesh[0] * mesh[1] * (mesh[2] / 2 + 1);
  ARG_CHECK(h, ldt == G);   // FFT batches read the rows in place
  if (nrows == 0) return ISDF_OK;
  if (batch > nrows) batch = nrows;
  double* cg = nullptr;
  int rc = get_coulG_half(h, mesh, a, 1.0, &cg);
  if (rc) return rc;
  double* V = (double*)isdf_ws(h, "coul_V", sizeof(double) * (size_t)batch * G);
  double2* Z = (double2*)isdf_ws(h, "coul_Z", sizeof(double2) * (size_t)batch * gc);
  if (!V || !Z) return ISDF_ERR_HIP;
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
                     a[2] * (a[3] * a[7] - a[4] * a[6]);
  const double w = fabs(det) / (double)G;
  for (int r = row0; r < row0 + nrows; r += batch) {
    const int nb = std::min(batch, row0 + nrows - r);
    rc = convolve_rows(h, d_theta + (int64_t)r * ldt, V, nb, mesh, cg, Z);
    if (rc) return rc;
    // W[r:r+nb, c0:] = w * V (nb x G) * Theta[c0:]^T ; with upper_only the columns left of the batch's
    // first row are skipped (filled later by isdf_symmetrize_upper)
    const int c0 = upper_only ? r : 0;
    rc = gemm_nt_f64(h, nb, P - c0, G, w, V, G, d_theta + (int64_t)c0 * ldt, ldt, 0.0,
                     d_W + (int64_t)r * ldw + c0, ldw);
    if (rc) return rc;
  }
  return ISDF_OK;
}

extern "C" int isdf_coulomb_potential(isdf_handle h, double* d_rho, int nset, int64_t ldrho,
                                      const int32_t mesh[3], const double a[9]) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_rho && mesh && a && nset > 0);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  const int64_t gc = (int64_t)mesh[0] * mesh[1] * (mesh[2] / 2 + 1);
  ARG_CHECK(h, ldrho == G);
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
                     a[2] * (a[3] * a[7] - a[4] * a[6]);
  double* cg = nullptr;
  int rc = get_coulG_half(h, mesh, a, fabs(det) / (double)G, &cg);   // weight vol/G folded in
  if (rc) return rc;
  double2* Z = (double2*)isdf_ws(h, "coul_Z", sizeof(double2) * (size_t)nset * gc);
  if (!Z) return ISDF_ERR_HIP;
  return convolve_rows(h, d_rho, d_rho, nset, mesh, cg, Z);
}

extern "C" int isdf_coulomb_rows(isdf_handle h, const double* d_in, int nrows, int64_t ld,
                                 const int32_t mesh[3], const double a[9], int batch, double* d_out,
                                 int64_t ldo) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_in && d_out && mesh && a && nrows >= 0 && batch > 0);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  const int64_t gc = (int64_t)mesh[0] * mesh[1] * (mesh[2] / 2 + 1);
  ARG_CHECK(h, ld == G && ldo == G);
  if (nrows == 0) return ISDF_OK;
  if (batch > nrows) batch = nrows;
  double* cg = nullptr;
  int rc = get_coulG_half(h, mesh, a, 1.0, &cg);
  if (rc) return rc;
  double2* Z = (double2*)isdf_ws(h, "coul_Z", sizeof(double2) * (size_t)batch * gc);
  if (!Z) return ISDF_ERR_HIP;
  for (int r = 0; r < nrows; r += batch) {
    const int nb = std::min(batch, nrows - r);
    rc = convolve_rows(h, d_in + (int64_t)r * ld, d_out + (int64_t)r * ldo, nb, mesh, cg, Z);
    if (rc) return rc;
  }
  return ISDF_OK;
}

// ---- spectral form of W: the kernel table the Gamma-point convolution uses, and the packed spectra ------
extern "C" int isdf_coulG_half(isdf_handle h, const int32_t mesh[3], const double a[9], double* d_out) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, mesh && a && d_out);
  double* cg = nullptr;
  int rc = get_coulG_half(h, mesh, a, 1.0, &cg);
  if (rc) return rc;
  const int64_t gc = (int64_t)mesh[0] * mesh[1] * (mesh[2] / 2 + 1);
  HIP_TRY(h, hipMemcpyAsync(d_out, cg, sizeof(double) * gc, hipMemcpyDeviceToDevice, h->stream));
  return ISDF_OK;
}

extern "C" int isdf_spectral_supported(isdf_handle h, const int32_t mesh[3], int batch, int* ok) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, mesh && ok && batch > 0);
  *ok = spectral_rows_own_supported(h, mesh, batch) ? 1 : 0;
  return ISDF_OK;
}

extern "C" int isdf_spectral_rows(isdf_handle h, const double* d_in, int nrows, int64_t ld, const int32_t mesh[3],
                                  const int32_t* d_idx, const double* d_scale, int npts, int batch, double* d_out, int64_t ldx) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_in && d_out && mesh && d_idx && d_scale && nrows >= 0 && batch > 0 && npts > 0);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  const int64_t gc = (int64_t)mesh[0] * mesh[1] * (mesh[2] / 2 + 1);
  ARG_CHECK(h, ld == G && npts <= gc);
  if (nrows == 0) return ISDF_OK;
  if (batch > nrows) batch = nrows;
  if (!spectral_rows_own_supported(h, mesh, batch))
    return isdf_fail(h, ISDF_ERR_ARG, "isdf_spectral_rows: mesh %d x %d x %d is not covered by the plane FFT", mesh[0], mesh[1], mesh[2]);
  double2* Z = (double2*)isdf_ws(h, "coul_Z", sizeof(double2) * (size_t)batch * gc);
  if (!Z) return ISDF_ERR_HIP;
  for (int r = 0; r < nrows; r += batch) {
    const int nb = std::min(batch, nrows - r);
    int rc = spectral_rows_own(h, d_in + (int64_t)r * ld, nb, mesh, d_idx, d_scale, npts, d_out + (int64_t)r * ldx, ldx, Z);
    if (rc) return rc;
  }
  return ISDF_OK;
}

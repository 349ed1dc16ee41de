// Multigrid J / LDA potential (the role of pyscf/pbc/dft/multigrid/multigrid.py:500-680,838-935,
// 1046-1150): the density of the sharp basis functions is collocated on the dense mesh, that of smoother ones on coarser
// level meshes; the level spectra are added into the dense mesh's spectrum at the matching frequencies, the potential
// spectrum is cut back to each level and integrated there.
//
// What lives here: the spectrum traffic between a level mesh and the dense mesh (half spectra of real fields throughout:
// D2Z / Z2D, (n0, n1, n2/2+1) complex per field), the rectangular density contraction rho = sum_(mu in A, nu in B) aoA D aoB,
// the Coulomb kernel on a spectrum, the LDA exchange kernel and a deterministic dot product.  The rectangular potential
// integral V = aoA (v .* aoB)^T is isdf_gemm_nt with its per-k scale.
//
// Frequencies follow numpy.fft.fftfreq like the reference's index lists (multigrid.py:669-673): index i of an n-point axis
// carries f = i for i < (n+1)/2, else i - n, and sits at index f (f >= 0) or N + f (f < 0) of the N-point dense axis.  Along
// the halved axis only f >= 0 is stored on both meshes.  The Nyquist entries of an even level mesh need care (see nyq_info):
// the reference keeps a non-Hermitian spectrum and the real part of the field; here the Hermitian equivalent is stored.
#include "common.h"

namespace {

constexpr int64_t RCHUNK = 32768;   // grid columns per pass of the density contraction

__device__ inline int dense_index(int i, int n, int N) {
  const int f = (i < (n + 1) / 2) ? i : i - n;
  return f >= 0 ? f : f + N;
}

// Nyquist entries of an even level mesh (index n/2 along an axis on which the dense mesh is finer - "proper" Nyquist below).
// The reference places such an entry v at dense frequency -n/2 ONLY (fftfreq labels it so, multigrid.py:669-673), leaves +n/2
// empty and takes .real of the transformed field at the end: the real field it keeps has the spectrum (R(f) + conj(R(-f)))/2,
// i.e. v/2 at f_L and conj(v)/2 at -f_L for an entry whose frequency vector f_L has a proper Nyquist component (every other entry
// meets its own Hermitian partner and keeps weight 1).  A half spectrum can only hold Hermitian fields, so that symmetrised
// form is what is stored: weight 1/2, at f_L when its z component is stored (>= 0), else as the conjugate at -f_L; in a
// self-conjugate z plane (z = 0, or the dense mesh's own Nyquist plane) both f_L and -f_L are stored and both are written; in an
// interior z plane the thread also writes for the level's unstored partner entry (see the kernel).
// The map stays injective: a +n/2 component is produced by these entries only.
struct NyqInfo { bool pnx, pny, pnz, any; };
__device__ inline NyqInfo nyq_info(int ix, int iy, int iz, int n0, int n1, int n2, int N0, int N1, int N2) {
  NyqInfo q;
  q.pnx = (n0 % 2 == 0) && ix == n0 / 2 && N0 > n0;
  q.pny = (n1 % 2 == 0) && iy == n1 / 2 && N1 > n1;
  q.pnz = (n2 % 2 == 0) && iz == n2 / 2 && N2 > n2;
  q.any = q.pnx || q.pny || q.pnz;
  return q;
}

// full[set][dense(ix,iy,iz)] (+)= scale * sub[set][ix,iy,iz]; the map is injective, so no two threads meet
__global__ void spectrum_embed_kernel(const double2* __restrict__ sub, int n0, int n1, int n2h, int n2, double2* __restrict__ full,
                                      int N0, int N1, int N2h, int N2, double scale, int accumulate) {
  const int64_t gc = (int64_t)n0 * n1 * n2h;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= gc) return;
  const int set = blockIdx.y;
  const int iz = (int)(idx % n2h);
  const int iy = (int)((idx / n2h) % n1);
  const int ix = (int)(idx / ((int64_t)n2h * n1));
  const NyqInfo q = nyq_info(ix, iy, iz, n0, n1, n2, N0, N1, N2);
  const int jx = dense_index(ix, n0, N0), jy = dense_index(iy, n1, N1);
  const int mx = (N0 - jx) % N0, my = (N1 - jy) % N1;                       // the dense indices of -fx, -fy
  const double2 v = sub[(int64_t)set * gc + idx];
  const double w = q.any ? 0.5 * scale : scale;
  double2* base = full + (int64_t)set * N0 * N1 * N2h;
  auto put = [&](int tx, int ty, int tz, double re, double im) {
    double2* p = base + ((int64_t)tx * N1 + ty) * N2h + tz;
    if (accumulate) { p->x += re; p->y += im; }
    else { p->x = re; p->y = im; }
  };
  if (q.pnz) {
    put(mx, my, iz, w * v.x, -w * v.y);                                      // f_L has z = -n2/2: stored as the conjugate at -f_L
  } else {
    put(jx, jy, iz, w * v.x, w * v.y);
    if (q.any) {
      const bool self_z = iz == 0 || ((n2 % 2 == 0) && iz == n2 / 2);       // (the second case: N2 == n2 here)
      if (self_z) {
        put(mx, my, iz, w * v.x, -w * v.y);                                  // -f_L lies in the same, self-conjugate, stored plane
      } else {
        // an interior z plane: the level's partner entry (-ix, -iy, -iz) is not stored on the level side, so this thread also
        // places ITS contribution conj(conj(v))/2 - at f_L with the proper Nyquist components flipped to +n/2
        put(q.pnx ? n0 / 2 : jx, q.pny ? n1 / 2 : jy, iz, w * v.x, w * v.y);
      }
    }
  }
}

// sub[set][ix,iy,iz] = scale * (the reference's level spectrum after its .real): for a Hermitian dense spectrum V that is
// V(f_L) for ordinary entries and (V(f_L) + V(f_L with its proper Nyquist components flipped to +n/2)) / 2 for Nyquist entries
// (multigrid.py:905-915 picks the fftfreq entries and keeps the real part of the level field)
__global__ void spectrum_restrict_kernel(const double2* __restrict__ full, int N0, int N1, int N2h, int N2, double2* __restrict__ sub,
                                         int n0, int n1, int n2h, int n2, double scale) {
  const int64_t gc = (int64_t)n0 * n1 * n2h;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= gc) return;
  const int set = blockIdx.y;
  const int iz = (int)(idx % n2h);
  const int iy = (int)((idx / n2h) % n1);
  const int ix = (int)(idx / ((int64_t)n2h * n1));
  const NyqInfo q = nyq_info(ix, iy, iz, n0, n1, n2, N0, N1, N2);
  const int jx = dense_index(ix, n0, N0), jy = dense_index(iy, n1, N1);
  const double2* base = full + (int64_t)set * N0 * N1 * N2h;
  auto get = [&](int tx, int ty, int tz) { return base[((int64_t)tx * N1 + ty) * N2h + tz]; };
  double2 r;
  if (!q.any) {
    r = get(jx, jy, iz);
  } else {
    const int fx = q.pnx ? n0 / 2 : jx, fy = q.pny ? n1 / 2 : jy;            // flipped components: +n/2
    double2 a, b = get(fx, fy, iz);                                          // (flipped z = +n2/2 = iz when pnz)
    if (q.pnz) {
      a = get((N0 - jx) % N0, (N1 - jy) % N1, iz);                           // V(fx, fy, -n2/2) = conj V(-fx, -fy, +n2/2)
      a.y = -a.y;
    } else {
      a = get(jx, jy, iz);
    }
    r = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
  }
  sub[(int64_t)set * gc + idx] = make_double2(scale * r.x, scale * r.y);
}

__global__ void spectrum_scale_kernel(double2* __restrict__ z, const double* __restrict__ table, int64_t gc) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= gc) return;
  double2* q = z + (int64_t)blockIdx.y * gc + idx;
  const double c = table[idx];
  q->x *= c; q->y *= c;
}

// rho[g] = sum_mu T[mu, g] * aoA[mu, g]
__global__ void rho_pair_reduce_kernel(const double* __restrict__ T, int64_t ldT, const double* __restrict__ aoA, int64_t ld, int nA,
                                       int64_t ng, double* __restrict__ rho) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  double s = 0.0;
#pragma unroll 4
  for (int mu = 0; mu < nA; ++mu) s = fma(T[(int64_t)mu * ldT + g], aoA[(int64_t)mu * ld + g], s);
  rho[g] = s;
}

// Slater exchange of a spin-unpolarised density: exc = -(3/4) (3/pi)^(1/3) rho^(1/3) per particle, vxc = (4/3) exc.
// Densities at or below 1e-24 (the noise floor of the collocation, negative ripples of the FFT) give zero.
__global__ void lda_exchange_kernel(const double* __restrict__ rho, int64_t n, double* __restrict__ exc, double* __restrict__ vxc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double r = rho[i];
  double e = 0.0;
  if (r > 1e-24) e = -0.75 * cbrt(3.0 / 3.14159265358979323846) * cbrt(r);
  exc[i] = e;
  vxc[i] = (4.0 / 3.0) * e;
}

// VWN5 correlation of a spin-unpolarised density (libxc LDA_C_VWN, the correlation of the reference's 'lda,vwn'; Vosko, Wilk,
// Nusair, Can. J. Phys. 58, 1200 (1980), eq. 4.4, paramagnetic parameters of fit V: A = 0.0310907, b = 3.72744, c = 12.9352,
// x0 = -0.10498), ADDED to exc / vxc:  eps_c = A { ln(x^2/X) + 2b/Q atan(Q/(2x+b)) - b x0/X(x0) [ ln((x-x0)^2/X) +
// 2(b+2 x0)/Q atan(Q/(2x+b)) ] },  x = sqrt(rs), X = x^2 + b x + c, Q = sqrt(4c - b^2);  v_c = eps_c - (x/6) d eps_c/dx.
__global__ void lda_vwn_add_kernel(const double* __restrict__ rho, int64_t n, double* __restrict__ exc, double* __restrict__ vxc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double r = rho[i];
  if (!(r > 1e-24)) return;
  const double A = 0.0310907, b = 3.72744, c = 12.9352, x0 = -0.10498;
  const double rs = cbrt(3.0 / (4.0 * 3.14159265358979323846 * r));
  const double x = sqrt(rs);
  const double X = x * x + b * x + c, X0 = x0 * x0 + b * x0 + c;
  const double Q = sqrt(4.0 * c - b * b);
  const double at = atan(Q / (2.0 * x + b));
  const double ec = A * (log(x * x / X) + 2.0 * b / Q * at - b * x0 / X0 * (log((x - x0) * (x - x0) / X) + 2.0 * (b + 2.0 * x0) / Q * at));
  const double den = Q * Q + (2.0 * x + b) * (2.0 * x + b);
  const double dec = A * (2.0 / x - (2.0 * x + b) / X - 4.0 * b / den
                          - b * x0 / X0 * (2.0 / (x - x0) - (2.0 * x + b) / X - 4.0 * (b + 2.0 * x0) / den));
  exc[i] += ec;
  vxc[i] += ec - x / 6.0 * dec;
}

// Becke-88 exchange of a spin-unpolarised density (libxc GGA_X_B88; Becke, PRA 38, 3098): per spin channel
// f(rho_s, g_s) = rho_s^(4/3) G(x), x = g_s / rho_s^(4/3), G = -C_x - beta x^2 / (1 + 6 beta x asinh x), C_x = (3/2)(3/4pi)^(1/3),
// beta = 0.0042; e(rho, grad rho) = 2 f(rho/2, |grad rho|/2).  Outputs: exc = e / rho, vrho = de/drho, and the vector
// w = de/d(grad rho) = 2 vsigma grad rho (what multiplies grad(phi_mu phi_nu) in the potential matrix).
__global__ void gga_b88_kernel(const double* __restrict__ rho, const double* __restrict__ grad, int64_t gstride, int64_t n,
                               double* __restrict__ exc, double* __restrict__ vrho, double* __restrict__ w, int64_t wstride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double r = rho[i];
  const double gx = grad[i], gy = grad[gstride + i], gz = grad[2 * gstride + i];
  double e = 0.0, vr = 0.0, wfac = 0.0;
  if (r > 1e-14) {
    const double beta = 0.0042;
    const double cx = 1.5 * cbrt(3.0 / (4.0 * 3.14159265358979323846));
    const double rs = 0.5 * r;
    const double r13 = cbrt(rs), r43 = rs * r13;
    const double gs = 0.5 * sqrt(gx * gx + gy * gy + gz * gz);
    const double x = gs / r43;
    const double as = asinh(x);
    const double D = 1.0 + 6.0 * beta * x * as;
    const double Dp = 6.0 * beta * (as + x / sqrt(1.0 + x * x));
    const double Gx = -cx - beta * x * x / D;
    const double Gp_over_x = -beta * (2.0 * D - x * Dp) / (D * D);          // G'(x) / x, finite at x = 0
    e = 2.0 * r43 * Gx / r;
    vr = (4.0 / 3.0) * r13 * (Gx - x * x * Gp_over_x);
    // de/d|grad rho| = 2 f_gs / 2 = G'(x);  w = G'(x) grad rho / |grad rho| = (G'/x) grad rho / (2 rho_s^(4/3))
    wfac = Gp_over_x / (2.0 * r43);
  }
  exc[i] = e;
  vrho[i] = vr;
  w[i] = wfac * gx;
  w[wstride + i] = wfac * gy;
  w[2 * wstride + i] = wfac * gz;
}

// second derivative of the Slater exchange energy density: f = d2(rho exc)/d rho2 = (4/9) C rho^(-2/3), C = -(3/4)(3/pi)^(1/3)
__global__ void lda_exchange_fxc_kernel(const double* __restrict__ rho, int64_t n, double* __restrict__ fxc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double r = rho[i];
  double f = 0.0;
  if (r > 1e-24) {
    const double c = cbrt(r);
    f = -(1.0 / 3.0) * cbrt(3.0 / 3.14159265358979323846) / (c * c);
  }
  fxc[i] = f;
}

// two-stage deterministic reduction: partial[b] = sum over block b's strided elements of x (* y)
__global__ void dot_partial_kernel(const double* __restrict__ x, const double* __restrict__ y, int64_t n, double* __restrict__ partial) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += y ? x[i] * y[i] : x[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ void dot_final_kernel(const double* __restrict__ partial, int nb, double* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

// coords[d][g] = sum_i f_i a[i][d], f_i = numpy.fft.fftfreq(n_i)[index_i]: the uniform grid in the order and with the folding of
// cell.get_uniform_grids (cell.py:874-898, wrap_around = True), structure of arrays
__global__ void uniform_grid_kernel(double* __restrict__ coords, int n0, int n1, int n2, const double a0, const double a1,
                                    const double a2, const double a3, const double a4, const double a5, const double a6,
                                    const double a7, const double a8) {
  const int64_t G = (int64_t)n0 * n1 * n2;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const int iz = (int)(g % n2), iy = (int)((g / n2) % n1), ix = (int)(g / ((int64_t)n2 * n1));
  const double fx = (double)((ix < (n0 + 1) / 2) ? ix : ix - n0) / (double)n0;
  const double fy = (double)((iy < (n1 + 1) / 2) ? iy : iy - n1) / (double)n1;
  const double fz = (double)((iz < (n2 + 1) / 2) ? iz : iz - n2) / (double)n2;
  coords[g] = fx * a0 + fy * a3 + fz * a6;
  coords[G + g] = fx * a1 + fy * a4 + fz * a7;
  coords[2 * G + g] = fx * a2 + fy * a5 + fz * a8;
}

bool mesh_fits(const int32_t sub[3], const int32_t full[3]) {
  for (int d = 0; d < 3; ++d)
    if (sub[d] <= 0 || sub[d] > full[d]) return false;
  return true;
}

}  // namespace

extern "C" int isdf_uniform_grid(isdf_handle h, const int32_t mesh[3], const double a[9], double* d_coords_soa) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, mesh && a && d_coords_soa && mesh[0] > 0 && mesh[1] > 0 && mesh[2] > 0);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  hipLaunchKernelGGL(uniform_grid_kernel, dim3((unsigned)cdiv(G, 256)), dim3(256), 0, h->stream, d_coords_soa, mesh[0], mesh[1],
                     mesh[2], a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_rho_pair(isdf_handle h, const double* d_aoA, int nA, const double* d_aoB, int nB, int64_t ng, int64_t ld,
                             const double* d_dm, int nset, double* d_rho, int64_t ldrho) {
  // rho[i, g] = sum_(mu < nA, nu < nB) aoA[mu, g] dm[i, mu, nu] aoB[nu, g]
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_aoA && d_aoB && d_dm && d_rho && nA > 0 && nB > 0 && ng > 0 && ld >= ng && nset > 0 && ldrho >= ng);
  double* T = (double*)isdf_ws(h, "mg_T", sizeof(double) * (size_t)nA * RCHUNK);
  if (!T) return ISDF_ERR_HIP;
  for (int i = 0; i < nset; ++i) {
    for (int64_t g0 = 0; g0 < ng; g0 += RCHUNK) {
      const int64_t nc = std::min(RCHUNK, ng - g0);
      int rc = gemm_rm(h, 'N', 'N', nA, nc, nB, 1.0, d_dm + (int64_t)i * nA * nB, nB, d_aoB + g0, ld, 0.0, T, RCHUNK);
      if (rc) return rc;
      ProfScope ps(h, "rho_pair_reduce_kernel[byte]", 16.0 * (double)nA * (double)nc);
      hipLaunchKernelGGL(rho_pair_reduce_kernel, dim3((unsigned)cdiv(nc, 256)), dim3(256), 0, h->stream, T, RCHUNK, d_aoA + g0, ld,
                         nA, nc, d_rho + (int64_t)i * ldrho + g0);
      KERNEL_CHECK(h);
    }
  }
  return ISDF_OK;
}

extern "C" int isdf_mg_embed_density(isdf_handle h, const double* d_field, int nset, const int32_t mesh_sub[3], double scale,
                                     double* d_spec, const int32_t mesh[3], int accumulate) {
  // spec[set] (+)= scale * embed(fft(field[set] on mesh_sub)) - half spectra, fields contiguous (nset, prod(mesh_sub))
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_field && d_spec && mesh_sub && mesh && nset > 0 && nset <= 65535 && mesh_fits(mesh_sub, mesh));
  const int n2h = mesh_sub[2] / 2 + 1, N2h = mesh[2] / 2 + 1;
  const int64_t gc = (int64_t)mesh_sub[0] * mesh_sub[1] * n2h;
  double2* Z = (double2*)isdf_ws(h, "mg_Z", sizeof(double2) * (size_t)nset * gc);
  if (!Z) return ISDF_ERR_HIP;
  FftPlan* plan = nullptr;
  int rc = isdf_get_plan(h, mesh_sub, nset, &plan);
  if (rc) return rc;
  const int64_t G = (int64_t)mesh_sub[0] * mesh_sub[1] * mesh_sub[2];
  ProfScope ps(h, "mg_d2z_embed[byte]", (8.0 * G + 16.0 * gc * 3) * nset, 2);
  FFT_TRY(h, hipfftExecD2Z(plan->fwd, (hipfftDoubleReal*)d_field, (hipfftDoubleComplex*)Z));
  hipLaunchKernelGGL(spectrum_embed_kernel, dim3((unsigned)cdiv(gc, 256), (unsigned)nset), dim3(256), 0, h->stream, Z, mesh_sub[0],
                     mesh_sub[1], n2h, mesh_sub[2], (double2*)d_spec, mesh[0], mesh[1], N2h, mesh[2], scale, accumulate);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_mg_restrict_potential(isdf_handle h, const double* d_spec, int nset, const int32_t mesh[3],
                                          const int32_t mesh_sub[3], double scale, double* d_field) {
  // field[set] = scale * ifft_unnormalised(restrict(spec[set]) to mesh_sub)   (Z2D; pass scale = 1 / prod(mesh_sub) for numpy's ifft)
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_field && d_spec && mesh_sub && mesh && nset > 0 && nset <= 65535 && mesh_fits(mesh_sub, mesh));
  const int n2h = mesh_sub[2] / 2 + 1, N2h = mesh[2] / 2 + 1;
  const int64_t gc = (int64_t)mesh_sub[0] * mesh_sub[1] * n2h;
  double2* Z = (double2*)isdf_ws(h, "mg_Z", sizeof(double2) * (size_t)nset * gc);
  if (!Z) return ISDF_ERR_HIP;
  FftPlan* plan = nullptr;
  int rc = isdf_get_plan(h, mesh_sub, nset, &plan);
  if (rc) return rc;
  const int64_t G = (int64_t)mesh_sub[0] * mesh_sub[1] * mesh_sub[2];
  ProfScope ps(h, "mg_restrict_z2d[byte]", (8.0 * G + 16.0 * gc * 3) * nset, 2);
  hipLaunchKernelGGL(spectrum_restrict_kernel, dim3((unsigned)cdiv(gc, 256), (unsigned)nset), dim3(256), 0, h->stream,
                     (const double2*)d_spec, mesh[0], mesh[1], N2h, mesh[2], Z, mesh_sub[0], mesh_sub[1], n2h, mesh_sub[2], scale);
  KERNEL_CHECK(h);
  FFT_TRY(h, hipfftExecZ2D(plan->bwd, (hipfftDoubleComplex*)Z, (hipfftDoubleReal*)d_field));
  return ISDF_OK;
}

extern "C" int isdf_mg_coulomb_kernel(isdf_handle h, double* d_spec, int nset, const int32_t mesh[3], const double a[9]) {
  // spec[set] *= coulG (the handle's kernel: 4 pi / G^2 with its range-separation / truncation state, G = 0 -> 0)
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_spec && mesh && a && nset > 0 && nset <= 65535 && mesh[0] > 0 && mesh[1] > 0 && mesh[2] > 0);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  const int64_t gc = (int64_t)mesh[0] * mesh[1] * (mesh[2] / 2 + 1);
  double* cg = nullptr;
  int rc = get_coulG_half(h, mesh, a, (double)G, &cg);   // the table carries 1/G by default; undo it
  if (rc) return rc;
  hipLaunchKernelGGL(spectrum_scale_kernel, dim3((unsigned)cdiv(gc, 256), (unsigned)nset), dim3(256), 0, h->stream, (double2*)d_spec,
                     cg, gc);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_lda_exchange(isdf_handle h, const double* d_rho, int64_t n, double* d_exc, double* d_vxc) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_rho && d_exc && d_vxc && n > 0);
  hipLaunchKernelGGL(lda_exchange_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream, d_rho, n, d_exc, d_vxc);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_lda_vwn_add(isdf_handle h, const double* d_rho, int64_t n, double* d_exc, double* d_vxc) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_rho && d_exc && d_vxc && n > 0);
  hipLaunchKernelGGL(lda_vwn_add_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream, d_rho, n, d_exc, d_vxc);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_gga_b88(isdf_handle h, const double* d_rho, const double* d_grad, int64_t gstride, int64_t n, double* d_exc,
                            double* d_vrho, double* d_w, int64_t wstride) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_rho && d_grad && d_exc && d_vrho && d_w && n > 0 && gstride >= n && wstride >= n);
  hipLaunchKernelGGL(gga_b88_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream, d_rho, d_grad, gstride, n, d_exc,
                     d_vrho, d_w, wstride);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_lda_exchange_fxc(isdf_handle h, const double* d_rho, int64_t n, double* d_fxc) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_rho && d_fxc && n > 0);
  hipLaunchKernelGGL(lda_exchange_fxc_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream, d_rho, n, d_fxc);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_dot(isdf_handle h, const double* d_x, const double* d_y, int64_t n, double* result) {
  // *result = sum_i x_i y_i (d_y NULL: sum_i x_i); fixed reduction order; synchronises the work stream
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_x && result && n > 0);
  const int nb = (int)std::min<int64_t>(cdiv(n, 256), 1024);
  double* part = (double*)isdf_ws(h, "dot_partial", sizeof(double) * 1025);
  if (!part) return ISDF_ERR_HIP;
  hipLaunchKernelGGL(dot_partial_kernel, dim3((unsigned)nb), dim3(256), 0, h->stream, d_x, d_y, n, part);
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, h->stream, part, nb, part + 1024);
  KERNEL_CHECK(h);
  HIP_TRY(h, hipMemcpyAsync(result, part + 1024, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return ISDF_OK;
}

// GTH pseudopotential pieces of FFTDF.get_pp on the device ("next" row).
//   local part:     vlocR = ifft( -sum_a SI_a(G) vloc_a(G) ).real            (pyscf/pbc/df/fft.py:72-86,
//                   pyscf/pbc/gto/pseudo/pp.py:58-93, pp_int.py:51-71)
//   non-local part: overlaps <projector | AO> accumulated over G chunks          (fft.py:88-140)
//                   SPG_ao[j, mu] = sum_G conj(SI_a(G)) p_j(G+k) * aoG[mu](G+k),
//                   aoG = analytic Fourier transform of the AOs / sqrt(vol) (libcint's ft_ao restated:
//                   FT[S_lm(r-R) e^{-a|r-R|^2}](q) = (-i)^l (pi/a)^{3/2} (2a)^{-l} S_lm(q) e^{-q^2/4a} e^{-iq.R})
// The small final contraction with the h_ij matrices is done by the caller.  Checker: oracle/pp.py,
// pinned to pyscf/pbc/df/test/test_fft.py:601-611.
#include "common.h"

namespace {

constexpr double PI = 3.14159265358979323846;
constexpr double FAC_S = 0.282094791773878143;
constexpr double FAC_P = 0.488602511902919921;
constexpr double D_XY = 1.0925484305920792;
constexpr double D_Z2_ZZ = 0.6307831305050401;
constexpr double D_Z2_XXYY = 0.31539156525252005;
constexpr double D_X2Y2 = 0.5462742152960396;

struct Lat { double b[9]; };

__device__ inline void gvec(int64_t idx, int n0, int n1, int n2, const Lat& r, double* g) {
  const int iz = (int)(idx % n2);
  const int iy = (int)((idx / n2) % n1);
  const int ix = (int)(idx / ((int64_t)n2 * n1));
  const double fx = (ix < (n0 + 1) / 2) ? ix : ix - n0;
  const double fy = (iy < (n1 + 1) / 2) ? iy : iy - n1;
  const double fz = (iz < (n2 + 1) / 2) ? iz : iz - n2;
  g[0] = fx * r.b[0] + fy * r.b[3] + fz * r.b[6];
  g[1] = fx * r.b[1] + fy * r.b[4] + fz * r.b[7];
  g[2] = fx * r.b[2] + fy * r.b[5] + fz * r.b[8];
}

__device__ inline void solid_harmonics(int l, double x, double y, double z, double* s) {
  if (l == 0) { s[0] = FAC_S; }
  else if (l == 1) { s[0] = FAC_P * x; s[1] = FAC_P * y; s[2] = FAC_P * z; }
  else {
    s[0] = D_XY * x * y; s[1] = D_XY * y * z;
    s[2] = D_Z2_ZZ * z * z - D_Z2_XXYY * (x * x + y * y);
    s[3] = D_XY * x * z; s[4] = D_X2Y2 * (x * x - y * y);
  }
}

// pp_par per atom: [has_pp, Z, rloc, nexp, C1, C2, C3, C4];  out[G] complex = -sum_a SI_a vloc_a
__global__ void vloc_G_kernel(const double* __restrict__ coords, const double* __restrict__ par, int natm, int n0,
                              int n1, int n2, Lat lat, double2* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t G = (int64_t)n0 * n1 * n2;
  if (idx >= G) return;
  double g[3];
  gvec(idx, n0, n1, n2, lat, g);
  const double g2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
  const double coul = (idx == 0) ? 0.0 : 4.0 * PI / g2;
  double re = 0.0, im = 0.0;
  for (int a = 0; a < natm; ++a) {
    const double* p = par + 8 * a;
    double v = p[1] * coul;
    if (p[0] != 0.0) {
      const double rloc = p[2];
      const double x = g2 * rloc * rloc;
      const double ex = exp(-0.5 * x);
      v *= ex;
      if (idx == 0) v = -2.0 * PI * p[1] * rloc * rloc;
      const int nexp = (int)p[3];
      double cf = 0.0;
      if (nexp >= 1) cf += p[4];
      if (nexp >= 2) cf += p[5] * (3.0 - x);
      if (nexp >= 3) cf += p[6] * (15.0 - 10.0 * x + x * x);
      if (nexp >= 4) cf += p[7] * (105.0 - 105.0 * x + 21.0 * x * x - x * x * x);
      v -= pow(2.0 * PI, 1.5) * rloc * rloc * rloc * ex * cf;
    }
    double s, c;
    sincos(-(g[0] * coords[3 * a] + g[1] * coords[3 * a + 1] + g[2] * coords[3 * a + 2]), &s, &c);
    re -= c * v;
    im -= s * v;
  }
  out[idx] = make_double2(re, im);
}

__global__ void real_part_scaled_kernel(const double2* __restrict__ z, double* __restrict__ out, int64_t n, double scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = z[i].x * scale;
}

struct FtShell { int atom, l, nprim, nctr, pexp, pcoef, ao0; };

// aoG[mu][g] (nao, chunk) complex for wave vectors q = G + k, g in [g0, g0 + nc)
__global__ void ft_ao_kernel(const FtShell* __restrict__ shells, int nbas, const double* __restrict__ env,
                             const double* __restrict__ coords, int n0, int n1, int n2, Lat lat, double kx, double ky,
                             double kz, int64_t g0, int64_t nc, double scale, double2* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nc) return;
  double q[3];
  gvec(g0 + t, n0, n1, n2, lat, q);
  q[0] += kx; q[1] += ky; q[2] += kz;
  const double q2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  for (int ib = 0; ib < nbas; ++ib) {
    const FtShell sh = shells[ib];
    const double* R = coords + 3 * sh.atom;
    double s, c;
    sincos(-(q[0] * R[0] + q[1] * R[1] + q[2] * R[2]), &s, &c);
    // (-i)^l * exp(-i q.R)
    double pr, pi;
    if (sh.l == 0) { pr = c; pi = s; }
    else if (sh.l == 1) { pr = s; pi = -c; }
    else { pr = -c; pi = -s; }
    double S[5];
    solid_harmonics(sh.l, q[0], q[1], q[2], S);
    const int deg = 2 * sh.l + 1;
    for (int k = 0; k < sh.nctr; ++k) {
      double rad = 0.0;
      for (int p = 0; p < sh.nprim; ++p) {
        const double e = env[sh.pexp + p];
        rad += env[sh.pcoef + k * sh.nprim + p] * pow(PI / e, 1.5) * pow(2.0 * e, -(double)sh.l) * exp(-q2 / (4.0 * e));
      }
      rad *= scale;
      for (int m = 0; m < deg; ++m) {
        const double v = rad * S[m];
        out[(int64_t)(sh.ao0 + k * deg + m) * nc + t] = make_double2(v * pr, v * pi);
      }
    }
  }
}

struct Proj { int atom, l, i, row0; double rl; };

__device__ inline double qli(double x, int l, int i) {
  const double x2 = x * x;
  if (l == 0) return i == 0 ? 4.0 * sqrt(2.0) : (i == 1 ? 8.0 * sqrt(2.0 / 15.0) * (3.0 - x2)
                                                         : 16.0 / 3.0 * sqrt(2.0 / 105.0) * (15.0 - 10.0 * x2 + x2 * x2));
  if (l == 1) return i == 0 ? 8.0 * sqrt(1.0 / 3.0) : (i == 1 ? 16.0 * sqrt(1.0 / 105.0) * (5.0 - x2)
                                                               : 32.0 / 3.0 * sqrt(1.0 / 1155.0) * (35.0 - 14.0 * x2 + x2 * x2));
  return i == 0 ? 8.0 * sqrt(2.0 / 15.0) : (i == 1 ? 16.0 / 3.0 * sqrt(2.0 / 105.0) * (7.0 - x2)
                                                   : 32.0 / 3.0 * sqrt(2.0 / 15015.0) * (63.0 - 18.0 * x2 + x2 * x2));
}

// SPG[row][g] (nrows, chunk) complex = conj(SI_a(G)) * projector(G + k)
__global__ void proj_kernel(const Proj* __restrict__ projs, int nproj, const double* __restrict__ coords, int n0, int n1,
                            int n2, Lat lat, double kx, double ky, double kz, int64_t g0, int64_t nc,
                            double2* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nc) return;
  double g[3];
  gvec(g0 + t, n0, n1, n2, lat, g);
  const double q[3] = {g[0] + kx, g[1] + ky, g[2] + kz};
  const double qn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  for (int j = 0; j < nproj; ++j) {
    const Proj p = projs[j];
    const double* R = coords + 3 * p.atom;
    double s, c;
    sincos(g[0] * R[0] + g[1] * R[1] + g[2] * R[2], &s, &c);     // conj(SI) = exp(+i G.R)
    double S[5];
    solid_harmonics(p.l, q[0], q[1], q[2], S);
    const double base = pow(p.rl, p.l + 1.5) * pow(PI, 1.25) * exp(-0.5 * p.rl * p.rl * qn * qn) * qli(qn * p.rl, p.l, p.i);
    for (int m = 0; m < 2 * p.l + 1; ++m) {
      const double v = base * S[m];
      out[(int64_t)(p.row0 + m) * nc + t] = make_double2(v * c, v * s);
    }
  }
}

}  // namespace

static void recip(const double a[9], Lat* lat, double* vol) {
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
                     a[2] * (a[3] * a[7] - a[4] * a[6]);
  const double tp = 2.0 * PI / det;
  const double b[9] = {tp * (a[4] * a[8] - a[5] * a[7]), tp * (a[5] * a[6] - a[3] * a[8]), tp * (a[3] * a[7] - a[4] * a[6]),
                       tp * (a[7] * a[2] - a[8] * a[1]), tp * (a[8] * a[0] - a[6] * a[2]), tp * (a[6] * a[1] - a[7] * a[0]),
                       tp * (a[1] * a[5] - a[2] * a[4]), tp * (a[2] * a[3] - a[0] * a[5]), tp * (a[0] * a[4] - a[1] * a[3])};
  for (int i = 0; i < 9; ++i) lat->b[i] = b[i];
  *vol = fabs(det);
}

extern "C" int isdf_pp_local_potential(isdf_handle h, int natm, const double* coords, const double* pp_par,
                                       const int32_t mesh[3], const double a[9], double* d_vlocR) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, natm > 0 && coords && pp_par && mesh && a && d_vlocR);
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  ARG_CHECK(h, G < 2147483647LL);
  double* d_tab = (double*)isdf_ws(h, "pp_tab", sizeof(double) * (size_t)natm * 11);
  double2* Z = (double2*)isdf_ws(h, "pp_Z", sizeof(double2) * (size_t)G);
  if (!d_tab || !Z) return ISDF_ERR_HIP;
  HIP_TRY(h, hipMemcpyAsync(d_tab, coords, sizeof(double) * 3 * natm, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_tab + 3 * natm, pp_par, sizeof(double) * 8 * natm, hipMemcpyHostToDevice, h->stream));
  Lat lat; double vol;
  recip(a, &lat, &vol);
  hipLaunchKernelGGL(vloc_G_kernel, dim3((unsigned)cdiv(G, 256)), dim3(256), 0, h->stream, d_tab, d_tab + 3 * natm, natm,
                     mesh[0], mesh[1], mesh[2], lat, Z);
  KERNEL_CHECK(h);
  // single complex inverse transform (plan cached under batch = 1)
  std::vector<int> key = {mesh[0], mesh[1], mesh[2], 1, -1};
  hipfftHandle plan;
  auto it = h->plans.find(key);
  if (it == h->plans.end()) {
    FftPlan p;
    int dims[3] = {mesh[0], mesh[1], mesh[2]};
    FFT_TRY(h, hipfftPlanMany(&p.fwd, 3, dims, nullptr, 1, (int)G, nullptr, 1, (int)G, HIPFFT_Z2Z, 1));
    p.bwd = 0;
    h->plans.emplace(key, p);
    plan = p.fwd;
  } else plan = it->second.fwd;
  FFT_TRY(h, hipfftSetStream(plan, h->stream));
  FFT_TRY(h, hipfftExecZ2Z(plan, (hipfftDoubleComplex*)Z, (hipfftDoubleComplex*)Z, HIPFFT_BACKWARD));
  hipLaunchKernelGGL(real_part_scaled_kernel, dim3((unsigned)cdiv(G, 256)), dim3(256), 0, h->stream, Z, d_vlocR, G,
                     1.0 / (double)G);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_pp_projector_overlaps(isdf_handle h, const int32_t* atm, int natm, const int32_t* bas, int nbas,
                                          const double* env, int nenv, const double* coords, const double kpt[3],
                                          const int32_t* proj_tab, const double* proj_rl, int nproj,
                                          const int32_t mesh[3], const double a[9], double* d_out) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, atm && bas && env && coords && kpt && proj_tab && proj_rl && mesh && a && d_out && nproj > 0);
  constexpr int BAS_SLOTS = 8, ATOM_OF = 0, ANG_OF = 1, NPRIM_OF = 2, NCTR_OF = 3, PTR_EXP = 5, PTR_COEFF = 6;
  const int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  std::vector<FtShell> shells(nbas);
  int nao = 0;
  for (int ib = 0; ib < nbas; ++ib) {
    const int32_t* b = bas + ib * BAS_SLOTS;
    if (b[ANG_OF] > 2) return isdf_fail(h, ISDF_ERR_ARG, "shell %d: only l<=2 is supported", ib);
    shells[ib] = {b[ATOM_OF], b[ANG_OF], b[NPRIM_OF], b[NCTR_OF], b[PTR_EXP], b[PTR_COEFF], nao};
    nao += (2 * b[ANG_OF] + 1) * b[NCTR_OF];
  }
  std::vector<Proj> projs(nproj);
  int nrows = 0;
  for (int j = 0; j < nproj; ++j) {
    const int l = proj_tab[3 * j + 1];
    ARG_CHECK(h, l >= 0 && l <= 2 && proj_tab[3 * j + 2] >= 0 && proj_tab[3 * j + 2] <= 2);
    projs[j] = {proj_tab[3 * j], l, proj_tab[3 * j + 2], nrows, proj_rl[j]};
    nrows += 2 * l + 1;
  }
  const int64_t CH = 16384;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t b_sh = al(sizeof(FtShell) * nbas), b_env = al(sizeof(double) * nenv), b_co = al(sizeof(double) * 3 * natm),
               b_pr = al(sizeof(Proj) * nproj);
  char* tab = (char*)isdf_ws(h, "pp_tables", b_sh + b_env + b_co + b_pr);
  double2* aoG = (double2*)isdf_ws(h, "pp_aoG", sizeof(double2) * (size_t)nao * CH);
  double2* SPG = (double2*)isdf_ws(h, "pp_SPG", sizeof(double2) * (size_t)nrows * CH);
  if (!tab || !aoG || !SPG) return ISDF_ERR_HIP;
  FtShell* d_sh = (FtShell*)tab;
  double* d_env = (double*)(tab + b_sh);
  double* d_co = (double*)(tab + b_sh + b_env);
  Proj* d_pr = (Proj*)(tab + b_sh + b_env + b_co);
  HIP_TRY(h, hipMemcpyAsync(d_sh, shells.data(), sizeof(FtShell) * nbas, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_env, env, sizeof(double) * nenv, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_co, coords, sizeof(double) * 3 * natm, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_pr, projs.data(), sizeof(Proj) * nproj, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  Lat lat; double vol;
  recip(a, &lat, &vol);
  const rocblas_double_complex one(1.0, 0.0), zero(0.0, 0.0);
  for (int64_t g0 = 0; g0 < G; g0 += CH) {
    const int64_t nc = std::min(CH, G - g0);
    hipLaunchKernelGGL(ft_ao_kernel, dim3((unsigned)cdiv(nc, 128)), dim3(128), 0, h->stream, d_sh, nbas, d_env, d_co,
                       mesh[0], mesh[1], mesh[2], lat, kpt[0], kpt[1], kpt[2], g0, nc, 1.0 / sqrt(vol), aoG);
    hipLaunchKernelGGL(proj_kernel, dim3((unsigned)cdiv(nc, 128)), dim3(128), 0, h->stream, d_pr, nproj, d_co, mesh[0],
                       mesh[1], mesh[2], lat, kpt[0], kpt[1], kpt[2], g0, nc, SPG);
    KERNEL_CHECK(h);
    // out (nrows, nao) += SPG (nrows, nc) * aoG^T (nc, nao): row-major C = A B^T  <=>  col-major C^T = B A^T
    const rocblas_double_complex beta = (g0 == 0) ? zero : one;
    BLAS_TRY(h, rocblas_zgemm(h->blas, rocblas_operation_transpose, rocblas_operation_none, nao, nrows, (int)nc, &one,
                              (const rocblas_double_complex*)aoG, (int)nc, (const rocblas_double_complex*)SPG, (int)nc,
                              &beta, (rocblas_double_complex*)d_out, nao));
  }
  return ISDF_OK;
}

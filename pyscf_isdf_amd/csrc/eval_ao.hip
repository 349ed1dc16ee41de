// S1: periodic AO collocation on gfx950 (Γ point, real spherical GTOs, l <= 3).
//
// One workgroup = 256 consecutive grid points x one atom.  The workgroup first culls the lattice
// translations against the bounding box of its points (LDS list), then every lane walks only the
// surviving images for each shell of the atom, accumulating the shell's AO values in registers,
// and writes them AO-major so that the 64 lanes of a wave store 512 contiguous bytes per AO row.
// Arithmetic follows pyscf/lib/gto/deriv1.c:31-58 (fac * sum_p c_p exp(-a_p r^2)) and :71-165
// (Cartesian monomials), with libcint's real-spherical d combination; truncation is per point
// (|r - R - T| < rcut[shell]) — see include/mi355_isdf.h.
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int NCMAX = 4;     // contractions per shell kept in registers

struct AtomDev {
  double x, y, z, rcut_max;
  int sh0, sh1;
};
struct ShellDev {
  int l, nprim, nctr, pexp, pcoef, ao0;
  double rcut2;
};

constexpr double FAC_S = 0.282094791773878143;
constexpr double FAC_P = 0.488602511902919921;
constexpr double D_XY = 1.0925484305920792;
constexpr double D_Z2_ZZ = 0.6307831305050401;
constexpr double D_Z2_XXYY = 0.31539156525252005;
constexpr double D_X2Y2 = 0.5462742152960396;
// f shells: libcint's real-spherical combination, m = -3 .. 3 (cart2sph table of libcint 6.1.1; the reference tree holds no
// fixture with f shells on this path: parity unpinned, orthonormality tested)
constexpr double F_3 = 0.5900435899266435;      // y (3x^2 - y^2), x (x^2 - 3y^2)
constexpr double F_2M = 2.890611442640554;      // x y z
constexpr double F_1 = 0.4570457994644658;      // y (4z^2 - x^2 - y^2), x (4z^2 - x^2 - y^2)
constexpr double F_0 = 0.3731763325901154;      // z (2z^2 - 3x^2 - 3y^2)
constexpr double F_2 = 1.445305721320277;       // z (x^2 - y^2)

// real-spherical angular polynomials of a shell (without the radial part)
template <int L>
__device__ inline void angular(double dx, double dy, double dz, double* __restrict__ ang) {
  if (L == 0) {
    ang[0] = 1.0;
  } else if (L == 1) {
    ang[0] = dx; ang[1] = dy; ang[2] = dz;
  } else if (L == 2) {
    ang[0] = D_XY * dx * dy;
    ang[1] = D_XY * dy * dz;
    ang[2] = D_Z2_ZZ * dz * dz - D_Z2_XXYY * (dx * dx + dy * dy);
    ang[3] = D_XY * dx * dz;
    ang[4] = D_X2Y2 * (dx * dx - dy * dy);
  } else {
    const double x2 = dx * dx, y2 = dy * dy, z2 = dz * dz;
    ang[0] = F_3 * dy * (3.0 * x2 - y2);
    ang[1] = F_2M * dx * dy * dz;
    ang[2] = F_1 * dy * (4.0 * z2 - x2 - y2);
    ang[3] = F_0 * dz * (2.0 * z2 - 3.0 * x2 - 3.0 * y2);
    ang[4] = F_1 * dx * (4.0 * z2 - x2 - y2);
    ang[5] = F_2 * dz * (x2 - y2);
    ang[6] = F_3 * dx * (x2 - 3.0 * y2);
  }
}

// the same with Cartesian gradients gang[m][x]
template <int L>
__device__ inline void angular_grad(const double* __restrict__ d, double* __restrict__ ang, double (*gang)[3]) {
  const double dx = d[0], dy = d[1], dz = d[2];
  angular<L>(dx, dy, dz, ang);
  if (L == 0) {
    gang[0][0] = gang[0][1] = gang[0][2] = 0.0;
  } else if (L == 1) {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int x = 0; x < 3; ++x) gang[m][x] = (m == x) ? 1.0 : 0.0;
  } else if (L == 2) {
    gang[0][0] = D_XY * dy;  gang[0][1] = D_XY * dx;  gang[0][2] = 0.0;
    gang[1][0] = 0.0;        gang[1][1] = D_XY * dz;  gang[1][2] = D_XY * dy;
    gang[2][0] = -2.0 * D_Z2_XXYY * dx;  gang[2][1] = -2.0 * D_Z2_XXYY * dy;  gang[2][2] = 2.0 * D_Z2_ZZ * dz;
    gang[3][0] = D_XY * dz;  gang[3][1] = 0.0;        gang[3][2] = D_XY * dx;
    gang[4][0] = 2.0 * D_X2Y2 * dx;  gang[4][1] = -2.0 * D_X2Y2 * dy;  gang[4][2] = 0.0;
  } else {
    const double x2 = dx * dx, y2 = dy * dy, z2 = dz * dz;
    gang[0][0] = 6.0 * F_3 * dx * dy;   gang[0][1] = 3.0 * F_3 * (x2 - y2);              gang[0][2] = 0.0;
    gang[1][0] = F_2M * dy * dz;        gang[1][1] = F_2M * dx * dz;                     gang[1][2] = F_2M * dx * dy;
    gang[2][0] = -2.0 * F_1 * dx * dy;  gang[2][1] = F_1 * (4.0 * z2 - x2 - 3.0 * y2);   gang[2][2] = 8.0 * F_1 * dy * dz;
    gang[3][0] = -6.0 * F_0 * dx * dz;  gang[3][1] = -6.0 * F_0 * dy * dz;               gang[3][2] = F_0 * (6.0 * z2 - 3.0 * x2 - 3.0 * y2);
    gang[4][0] = F_1 * (4.0 * z2 - 3.0 * x2 - y2);  gang[4][1] = -2.0 * F_1 * dx * dy;   gang[4][2] = 8.0 * F_1 * dx * dz;
    gang[5][0] = 2.0 * F_2 * dx * dz;   gang[5][1] = -2.0 * F_2 * dy * dz;               gang[5][2] = F_2 * (x2 - y2);
    gang[6][0] = 3.0 * F_3 * (x2 - y2); gang[6][1] = -6.0 * F_3 * dx * dy;               gang[6][2] = 0.0;
  }
}

__device__ inline double wave_min(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
__device__ inline double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

template <int L>
__device__ inline void shell_eval(const ShellDev sh, const double* __restrict__ env,
                                  const double* __restrict__ Ls, const int* __restrict__ img_list,
                                  int nlist, double px, double py, double pz, double ax, double ay,
                                  double az, bool valid, double* __restrict__ ao, int64_t ld,
                                  int64_t g) {
  constexpr int DEG = 2 * L + 1;
  double acc[NCMAX][DEG];
#pragma unroll
  for (int c = 0; c < NCMAX; ++c)
#pragma unroll
    for (int m = 0; m < DEG; ++m) acc[c][m] = 0.0;
  const double fac = (L == 0) ? FAC_S : (L == 1 ? FAC_P : 1.0);
  const double* __restrict__ es = env + sh.pexp;
  const double* __restrict__ cs = env + sh.pcoef;
  for (int i = 0; i < nlist; ++i) {
    const int iL = img_list[i];
    const double dx = px - (ax + Ls[3 * iL + 0]);
    const double dy = py - (ay + Ls[3 * iL + 1]);
    const double dz = pz - (az + Ls[3 * iL + 2]);
    const double rr = dx * dx + dy * dy + dz * dz;
    if (rr < sh.rcut2) {
      double rad[NCMAX];
#pragma unroll
      for (int c = 0; c < NCMAX; ++c) rad[c] = 0.0;
      for (int p = 0; p < sh.nprim; ++p) {
        const double e = exp(-es[p] * rr) * fac;
#pragma unroll
        for (int c = 0; c < NCMAX; ++c)
          if (c < sh.nctr) rad[c] += cs[c * sh.nprim + p] * e;
      }
      double ang[DEG];
      angular<L>(dx, dy, dz, ang);
#pragma unroll
      for (int c = 0; c < NCMAX; ++c)
#pragma unroll
        for (int m = 0; m < DEG; ++m) acc[c][m] += rad[c] * ang[m];
    }
  }
  if (valid) {
#pragma unroll
    for (int c = 0; c < NCMAX; ++c)
      if (c < sh.nctr) {
#pragma unroll
        for (int m = 0; m < DEG; ++m) ao[(int64_t)(sh.ao0 + c * DEG + m) * ld + g] = acc[c][m];
      }
  }
}

__global__ __launch_bounds__(TPB) void eval_ao_kernel(
    const AtomDev* __restrict__ atoms, const ShellDev* __restrict__ shells,
    const double* __restrict__ env, const double* __restrict__ Ls, int nimgs,
    const double* __restrict__ coords, int64_t ngrids, double* __restrict__ ao, int64_t ld) {
  extern __shared__ int img_list[];      // nimgs ints
  __shared__ double red[6][TPB / 64];
  __shared__ int wcnt[TPB / 64];
  const int tid = threadIdx.x;
  const int64_t g = (int64_t)blockIdx.x * TPB + tid;
  const bool valid = g < ngrids;
  const int64_t gc = valid ? g : (int64_t)blockIdx.x * TPB;   // clamp to the block's first point
  const double px = coords[gc], py = coords[ngrids + gc], pz = coords[2 * ngrids + gc];
  const AtomDev at = atoms[blockIdx.y];

  // bounding box of the workgroup's points
  double lo[3] = {wave_min(px), wave_min(py), wave_min(pz)};
  double hi[3] = {wave_max(px), wave_max(py), wave_max(pz)};
  const int w = tid >> 6;
  if ((tid & 63) == 0) {
    for (int k = 0; k < 3; ++k) { red[k][w] = lo[k]; red[3 + k][w] = hi[k]; }
  }
  __syncthreads();
  for (int k = 0; k < 3; ++k) {
    lo[k] = red[k][0]; hi[k] = red[3 + k][0];
    for (int ww = 1; ww < TPB / 64; ++ww) {
      lo[k] = fmin(lo[k], red[k][ww]);
      hi[k] = fmax(hi[k], red[3 + k][ww]);
    }
  }
  // cull images: keep T when dist(R + T, box) < rcut_max (conservative; the exact test is per
  // point).  Ordered compaction (ballot + prefix) keeps the list in ascending image index, so the
  // floating-point accumulation order over images is fixed and equals the oracle's.
  const double rc2 = at.rcut_max * at.rcut_max;
  const int lane = tid & 63;
  int nlist = 0;
  for (int base = 0; base < nimgs; base += TPB) {
    const int i = base + tid;
    bool keep = false;
    if (i < nimgs) {
      const double c[3] = {at.x + Ls[3 * i], at.y + Ls[3 * i + 1], at.z + Ls[3 * i + 2]};
      double d2 = 0.0;
      for (int k = 0; k < 3; ++k) {
        const double d = fmax(fmax(lo[k] - c[k], c[k] - hi[k]), 0.0);
        d2 += d * d;
      }
      keep = d2 < rc2;
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) wcnt[w] = __popcll(mask);
    __syncthreads();
    int off = nlist;
    for (int ww = 0; ww < w; ++ww) off += wcnt[ww];
    if (keep) img_list[off + __popcll(mask & ((1ull << lane) - 1ull))] = i;
    for (int ww = 0; ww < TPB / 64; ++ww) nlist += wcnt[ww];
    __syncthreads();
  }

  for (int s = at.sh0; s < at.sh1; ++s) {
    const ShellDev sh = shells[s];
    switch (sh.l) {
      case 0: shell_eval<0>(sh, env, Ls, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, ao, ld, g); break;
      case 1: shell_eval<1>(sh, env, Ls, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, ao, ld, g); break;
      case 2: shell_eval<2>(sh, env, Ls, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, ao, ld, g); break;
      default: shell_eval<3>(sh, env, Ls, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, ao, ld, g); break;
    }
  }
}

// k-point variant: every image contributes with the Bloch factor exp(i k.T) (table per image), the
// result is multiplied by exp(-i k.r) to give the lattice-periodic part u^k = exp(-i k.r) phi^k
// (periodic=1) or left as phi^k (periodic=0).  Output: real and imaginary planes.
template <int L>
__device__ inline void shell_eval_k(const ShellDev sh, const double* __restrict__ env,
                                    const double* __restrict__ Ls, const double* __restrict__ phT,
                                    const int* __restrict__ img_list, int nlist, double px, double py,
                                    double pz, double ax, double ay, double az, bool valid, double pr,
                                    double pi, double* __restrict__ out_re, double* __restrict__ out_im,
                                    int64_t ld, int64_t g) {
  constexpr int DEG = 2 * L + 1;
  double accr[NCMAX][DEG], acci[NCMAX][DEG];
#pragma unroll
  for (int c = 0; c < NCMAX; ++c)
#pragma unroll
    for (int m = 0; m < DEG; ++m) { accr[c][m] = 0.0; acci[c][m] = 0.0; }
  const double fac = (L == 0) ? FAC_S : (L == 1 ? FAC_P : 1.0);
  const double* __restrict__ es = env + sh.pexp;
  const double* __restrict__ cs = env + sh.pcoef;
  for (int i = 0; i < nlist; ++i) {
    const int iL = img_list[i];
    const double dx = px - (ax + Ls[3 * iL + 0]);
    const double dy = py - (ay + Ls[3 * iL + 1]);
    const double dz = pz - (az + Ls[3 * iL + 2]);
    const double rr = dx * dx + dy * dy + dz * dz;
    if (rr < sh.rcut2) {
      const double tr = phT[2 * iL], ti = phT[2 * iL + 1];
      double rad[NCMAX];
#pragma unroll
      for (int c = 0; c < NCMAX; ++c) rad[c] = 0.0;
      for (int p = 0; p < sh.nprim; ++p) {
        const double e = exp(-es[p] * rr) * fac;
#pragma unroll
        for (int c = 0; c < NCMAX; ++c)
          if (c < sh.nctr) rad[c] += cs[c * sh.nprim + p] * e;
      }
      double ang[DEG];
      angular<L>(dx, dy, dz, ang);
#pragma unroll
      for (int c = 0; c < NCMAX; ++c)
#pragma unroll
        for (int m = 0; m < DEG; ++m) {
          const double v = rad[c] * ang[m];
          accr[c][m] += v * tr;
          acci[c][m] += v * ti;
        }
    }
  }
  if (valid) {
#pragma unroll
    for (int c = 0; c < NCMAX; ++c)
      if (c < sh.nctr) {
#pragma unroll
        for (int m = 0; m < DEG; ++m) {
          const int64_t off = (int64_t)(sh.ao0 + c * DEG + m) * ld + g;
          out_re[off] = accr[c][m] * pr - acci[c][m] * pi;
          out_im[off] = accr[c][m] * pi + acci[c][m] * pr;
        }
      }
  }
}

__global__ __launch_bounds__(TPB) void eval_ao_k_kernel(
    const AtomDev* __restrict__ atoms, const ShellDev* __restrict__ shells,
    const double* __restrict__ env, const double* __restrict__ Ls, const double* __restrict__ phT,
    int nimgs, double kx, double ky, double kz, int periodic, const double* __restrict__ coords,
    int64_t ngrids, double* __restrict__ out_re, double* __restrict__ out_im, int64_t ld) {
  extern __shared__ int img_list[];
  __shared__ double red[6][TPB / 64];
  __shared__ int wcnt[TPB / 64];
  const int tid = threadIdx.x;
  const int64_t g = (int64_t)blockIdx.x * TPB + tid;
  const bool valid = g < ngrids;
  const int64_t gc = valid ? g : (int64_t)blockIdx.x * TPB;
  const double px = coords[gc], py = coords[ngrids + gc], pz = coords[2 * ngrids + gc];
  const AtomDev at = atoms[blockIdx.y];
  double lo[3] = {wave_min(px), wave_min(py), wave_min(pz)};
  double hi[3] = {wave_max(px), wave_max(py), wave_max(pz)};
  const int w = tid >> 6;
  if ((tid & 63) == 0) {
    for (int k = 0; k < 3; ++k) { red[k][w] = lo[k]; red[3 + k][w] = hi[k]; }
  }
  __syncthreads();
  for (int k = 0; k < 3; ++k) {
    lo[k] = red[k][0]; hi[k] = red[3 + k][0];
    for (int ww = 1; ww < TPB / 64; ++ww) {
      lo[k] = fmin(lo[k], red[k][ww]);
      hi[k] = fmax(hi[k], red[3 + k][ww]);
    }
  }
  const double rc2 = at.rcut_max * at.rcut_max;
  const int lane = tid & 63;
  int nlist = 0;
  for (int base = 0; base < nimgs; base += TPB) {
    const int i = base + tid;
    bool keep = false;
    if (i < nimgs) {
      const double c[3] = {at.x + Ls[3 * i], at.y + Ls[3 * i + 1], at.z + Ls[3 * i + 2]};
      double d2 = 0.0;
      for (int k = 0; k < 3; ++k) {
        const double d = fmax(fmax(lo[k] - c[k], c[k] - hi[k]), 0.0);
        d2 += d * d;
      }
      keep = d2 < rc2;
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) wcnt[w] = __popcll(mask);
    __syncthreads();
    int off = nlist;
    for (int ww = 0; ww < w; ++ww) off += wcnt[ww];
    if (keep) img_list[off + __popcll(mask & ((1ull << lane) - 1ull))] = i;
    for (int ww = 0; ww < TPB / 64; ++ww) nlist += wcnt[ww];
    __syncthreads();
  }
  // exp(-i k.r) for the periodic part (1 otherwise)
  double pr = 1.0, pi = 0.0;
  if (periodic) {
    const double kr = kx * px + ky * py + kz * pz;
    sincos(-kr, &pi, &pr);
  }
  for (int s = at.sh0; s < at.sh1; ++s) {
    const ShellDev sh = shells[s];
    switch (sh.l) {
      case 0: shell_eval_k<0>(sh, env, Ls, phT, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, pr, pi, out_re, out_im, ld, g); break;
      case 1: shell_eval_k<1>(sh, env, Ls, phT, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, pr, pi, out_re, out_im, ld, g); break;
      case 2: shell_eval_k<2>(sh, env, Ls, phT, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, pr, pi, out_re, out_im, ld, g); break;
      default: shell_eval_k<3>(sh, env, Ls, phT, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, pr, pi, out_re, out_im, ld, g); break;
    }
  }
}

// Values and Cartesian first derivatives (GGA densities and potentials; numint.eval_ao(deriv=1), pyscf/lib/gto/deriv1.c:60-69 for
// the radial part): phi = fac ang(d) R(r^2), R = sum_p c_p exp(-a_p r^2)  ->  grad phi = grad(ang) R - 2 d ang R1,
// R1 = sum_p c_p a_p exp(-a_p r^2).  Four output planes (value, d/dx, d/dy, d/dz), each AO-major like eval_ao_kernel's.
template <int L>
__device__ inline void shell_eval_d1(const ShellDev sh, const double* __restrict__ env, const double* __restrict__ Ls,
                                     const int* __restrict__ img_list, int nlist, double px, double py, double pz, double ax,
                                     double ay, double az, bool valid, double* __restrict__ ao, int64_t ld, int64_t plane,
                                     int64_t g) {
  constexpr int DEG = 2 * L + 1;
  double acc[4][NCMAX][DEG];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int c = 0; c < NCMAX; ++c)
#pragma unroll
      for (int m = 0; m < DEG; ++m) acc[x][c][m] = 0.0;
  const double fac = (L == 0) ? FAC_S : (L == 1 ? FAC_P : 1.0);
  const double* __restrict__ es = env + sh.pexp;
  const double* __restrict__ cs = env + sh.pcoef;
  for (int i = 0; i < nlist; ++i) {
    const int iL = img_list[i];
    const double d[3] = {px - (ax + Ls[3 * iL + 0]), py - (ay + Ls[3 * iL + 1]), pz - (az + Ls[3 * iL + 2])};
    const double rr = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (rr < sh.rcut2) {
      double rad[NCMAX], rad1[NCMAX];
#pragma unroll
      for (int c = 0; c < NCMAX; ++c) { rad[c] = 0.0; rad1[c] = 0.0; }
      for (int p = 0; p < sh.nprim; ++p) {
        const double e = exp(-es[p] * rr) * fac;
#pragma unroll
        for (int c = 0; c < NCMAX; ++c)
          if (c < sh.nctr) {
            rad[c] += cs[c * sh.nprim + p] * e;
            rad1[c] += cs[c * sh.nprim + p] * es[p] * e;
          }
      }
      double ang[DEG], gang[DEG][3];
      angular_grad<L>(d, ang, gang);
#pragma unroll
      for (int c = 0; c < NCMAX; ++c)
#pragma unroll
        for (int m = 0; m < DEG; ++m) {
          acc[0][c][m] += rad[c] * ang[m];
#pragma unroll
          for (int x = 0; x < 3; ++x) acc[1 + x][c][m] += gang[m][x] * rad[c] - 2.0 * d[x] * ang[m] * rad1[c];
        }
    }
  }
  if (valid) {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int c = 0; c < NCMAX; ++c)
        if (c < sh.nctr) {
#pragma unroll
          for (int m = 0; m < DEG; ++m) ao[x * plane + (int64_t)(sh.ao0 + c * DEG + m) * ld + g] = acc[x][c][m];
        }
  }
}

// the workgroup's image list (see eval_ao_kernel): ordered compaction of the translations whose atom image can reach the
// bounding box of the workgroup's points; returns the list length
__device__ inline int cull_images(const AtomDev at, const double* __restrict__ Ls, int nimgs, double px, double py, double pz,
                                  int* __restrict__ img_list, double (*red)[TPB / 64], int* __restrict__ wcnt) {
  const int tid = threadIdx.x;
  double lo[3] = {wave_min(px), wave_min(py), wave_min(pz)};
  double hi[3] = {wave_max(px), wave_max(py), wave_max(pz)};
  const int w = tid >> 6;
  if ((tid & 63) == 0) {
    for (int k = 0; k < 3; ++k) { red[k][w] = lo[k]; red[3 + k][w] = hi[k]; }
  }
  __syncthreads();
  for (int k = 0; k < 3; ++k) {
    lo[k] = red[k][0]; hi[k] = red[3 + k][0];
    for (int ww = 1; ww < TPB / 64; ++ww) {
      lo[k] = fmin(lo[k], red[k][ww]);
      hi[k] = fmax(hi[k], red[3 + k][ww]);
    }
  }
  const double rc2 = at.rcut_max * at.rcut_max;
  const int lane = tid & 63;
  int nlist = 0;
  for (int base = 0; base < nimgs; base += TPB) {
    const int i = base + tid;
    bool keep = false;
    if (i < nimgs) {
      const double c[3] = {at.x + Ls[3 * i], at.y + Ls[3 * i + 1], at.z + Ls[3 * i + 2]};
      double d2 = 0.0;
      for (int k = 0; k < 3; ++k) {
        const double dd = fmax(fmax(lo[k] - c[k], c[k] - hi[k]), 0.0);
        d2 += dd * dd;
      }
      keep = d2 < rc2;
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) wcnt[w] = __popcll(mask);
    __syncthreads();
    int off = nlist;
    for (int ww = 0; ww < w; ++ww) off += wcnt[ww];
    if (keep) img_list[off + __popcll(mask & ((1ull << lane) - 1ull))] = i;
    for (int ww = 0; ww < TPB / 64; ++ww) nlist += wcnt[ww];
    __syncthreads();
  }
  return nlist;
}

__global__ __launch_bounds__(TPB) void eval_ao_deriv1_kernel(
    const AtomDev* __restrict__ atoms, const ShellDev* __restrict__ shells, const double* __restrict__ env,
    const double* __restrict__ Ls, int nimgs, const double* __restrict__ coords, int64_t ngrids, double* __restrict__ ao,
    int64_t ld, int64_t plane) {
  extern __shared__ int img_list[];
  __shared__ double red[6][TPB / 64];
  __shared__ int wcnt[TPB / 64];
  const int64_t g = (int64_t)blockIdx.x * TPB + threadIdx.x;
  const bool valid = g < ngrids;
  const int64_t gc = valid ? g : (int64_t)blockIdx.x * TPB;
  const double px = coords[gc], py = coords[ngrids + gc], pz = coords[2 * ngrids + gc];
  const AtomDev at = atoms[blockIdx.y];
  const int nlist = cull_images(at, Ls, nimgs, px, py, pz, img_list, red, wcnt);
  for (int s = at.sh0; s < at.sh1; ++s) {
    const ShellDev sh = shells[s];
    switch (sh.l) {
      case 0: shell_eval_d1<0>(sh, env, Ls, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, ao, ld, plane, g); break;
      case 1: shell_eval_d1<1>(sh, env, Ls, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, ao, ld, plane, g); break;
      case 2: shell_eval_d1<2>(sh, env, Ls, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, ao, ld, plane, g); break;
      default: shell_eval_d1<3>(sh, env, Ls, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, ao, ld, plane, g); break;
    }
  }
}

// k-point form of shell_eval_d1, one component X (0 value, 1..3 d/dx, d/dy, d/dz) per call so that the accumulators stay in
// registers: Bloch sums sum_T exp(i k.T) (.)(r - T) of the value or of one derivative, times exp(-i k.r) when the periodic part is
// asked for (then the derivative is that of the Bloch function times the phase, NOT the derivative of the periodic part - callers
// that differentiate pair products conj(u_i) u_j never see the difference: the phases cancel in the product and in its gradient).
template <int L, int X>
__device__ inline void shell_eval_k_d1(const ShellDev sh, const double* __restrict__ env, const double* __restrict__ Ls,
                                       const double* __restrict__ phT, const int* __restrict__ img_list, int nlist, double px,
                                       double py, double pz, double ax, double ay, double az, bool valid, double pr, double pi,
                                       double* __restrict__ out_re, double* __restrict__ out_im, int64_t ld, int64_t g) {
  constexpr int DEG = 2 * L + 1;
  double accr[NCMAX][DEG], acci[NCMAX][DEG];
#pragma unroll
  for (int c = 0; c < NCMAX; ++c)
#pragma unroll
    for (int m = 0; m < DEG; ++m) { accr[c][m] = 0.0; acci[c][m] = 0.0; }
  const double fac = (L == 0) ? FAC_S : (L == 1 ? FAC_P : 1.0);
  const double* __restrict__ es = env + sh.pexp;
  const double* __restrict__ cs = env + sh.pcoef;
  for (int i = 0; i < nlist; ++i) {
    const int iL = img_list[i];
    const double d[3] = {px - (ax + Ls[3 * iL + 0]), py - (ay + Ls[3 * iL + 1]), pz - (az + Ls[3 * iL + 2])};
    const double rr = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (rr < sh.rcut2) {
      const double tr = phT[2 * iL], ti = phT[2 * iL + 1];
      double rad[NCMAX], rad1[NCMAX];
#pragma unroll
      for (int c = 0; c < NCMAX; ++c) { rad[c] = 0.0; rad1[c] = 0.0; }
      for (int p = 0; p < sh.nprim; ++p) {
        const double e = exp(-es[p] * rr) * fac;
#pragma unroll
        for (int c = 0; c < NCMAX; ++c)
          if (c < sh.nctr) {
            rad[c] += cs[c * sh.nprim + p] * e;
            if (X > 0) rad1[c] += cs[c * sh.nprim + p] * es[p] * e;
          }
      }
      double ang[DEG], gang[DEG], g3[DEG][3];
      angular_grad<L>(d, ang, g3);                  // (the two unused gradient components are dead code after inlining)
#pragma unroll
      for (int m = 0; m < DEG; ++m) gang[m] = g3[m][X > 0 ? X - 1 : 0];
#pragma unroll
      for (int c = 0; c < NCMAX; ++c)
#pragma unroll
        for (int m = 0; m < DEG; ++m) {
          const double v = (X == 0) ? rad[c] * ang[m] : gang[m] * rad[c] - 2.0 * d[X > 0 ? X - 1 : 0] * ang[m] * rad1[c];
          accr[c][m] += v * tr;
          acci[c][m] += v * ti;
        }
    }
  }
  if (valid) {
#pragma unroll
    for (int c = 0; c < NCMAX; ++c)
      if (c < sh.nctr) {
#pragma unroll
        for (int m = 0; m < DEG; ++m) {
          const int64_t off = (int64_t)(sh.ao0 + c * DEG + m) * ld + g;
          out_re[off] = accr[c][m] * pr - acci[c][m] * pi;
          out_im[off] = accr[c][m] * pi + acci[c][m] * pr;
        }
      }
  }
}

template <int L>
__device__ inline void shell_eval_k_d1_all(const ShellDev sh, const double* __restrict__ env, const double* __restrict__ Ls,
                                           const double* __restrict__ phT, const int* __restrict__ img_list, int nlist, double px,
                                           double py, double pz, double ax, double ay, double az, bool valid, double pr, double pi,
                                           double* __restrict__ out_re, double* __restrict__ out_im, int64_t ld, int64_t plane,
                                           int64_t g) {
  shell_eval_k_d1<L, 0>(sh, env, Ls, phT, img_list, nlist, px, py, pz, ax, ay, az, valid, pr, pi, out_re, out_im, ld, g);
  shell_eval_k_d1<L, 1>(sh, env, Ls, phT, img_list, nlist, px, py, pz, ax, ay, az, valid, pr, pi, out_re + plane, out_im + plane, ld, g);
  shell_eval_k_d1<L, 2>(sh, env, Ls, phT, img_list, nlist, px, py, pz, ax, ay, az, valid, pr, pi, out_re + 2 * plane, out_im + 2 * plane, ld, g);
  shell_eval_k_d1<L, 3>(sh, env, Ls, phT, img_list, nlist, px, py, pz, ax, ay, az, valid, pr, pi, out_re + 3 * plane, out_im + 3 * plane, ld, g);
}

__global__ __launch_bounds__(TPB) void eval_ao_k_deriv1_kernel(
    const AtomDev* __restrict__ atoms, const ShellDev* __restrict__ shells, const double* __restrict__ env,
    const double* __restrict__ Ls, const double* __restrict__ phT, int nimgs, double kx, double ky, double kz, int periodic,
    const double* __restrict__ coords, int64_t ngrids, double* __restrict__ out_re, double* __restrict__ out_im, int64_t ld,
    int64_t plane) {
  extern __shared__ int img_list[];
  __shared__ double red[6][TPB / 64];
  __shared__ int wcnt[TPB / 64];
  const int64_t g = (int64_t)blockIdx.x * TPB + threadIdx.x;
  const bool valid = g < ngrids;
  const int64_t gc = valid ? g : (int64_t)blockIdx.x * TPB;
  const double px = coords[gc], py = coords[ngrids + gc], pz = coords[2 * ngrids + gc];
  const AtomDev at = atoms[blockIdx.y];
  const int nlist = cull_images(at, Ls, nimgs, px, py, pz, img_list, red, wcnt);
  double pr = 1.0, pi = 0.0;
  if (periodic) {
    const double kr = kx * px + ky * py + kz * pz;
    sincos(-kr, &pi, &pr);
  }
  for (int s = at.sh0; s < at.sh1; ++s) {
    const ShellDev sh = shells[s];
    switch (sh.l) {
      case 0: shell_eval_k_d1_all<0>(sh, env, Ls, phT, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, pr, pi, out_re, out_im, ld, plane, g); break;
      case 1: shell_eval_k_d1_all<1>(sh, env, Ls, phT, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, pr, pi, out_re, out_im, ld, plane, g); break;
      case 2: shell_eval_k_d1_all<2>(sh, env, Ls, phT, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, pr, pi, out_re, out_im, ld, plane, g); break;
      default: shell_eval_k_d1_all<3>(sh, env, Ls, phT, img_list, nlist, px, py, pz, at.x, at.y, at.z, valid, pr, pi, out_re, out_im, ld, plane, g); break;
    }
  }
}

__global__ void gather_cols_kernel(const double* __restrict__ src, int64_t ld_src,
                                   const int64_t* __restrict__ idx, int64_t n,
                                   double* __restrict__ dst, int64_t ld_dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = blockIdx.y;
  dst[row * ld_dst + i] = src[row * ld_src + idx[i]];
}

}  // namespace

namespace {
struct AoTables {
  AtomDev* d_atoms; ShellDev* d_shells; double* d_env; double* d_Ls; double* d_phT; int nao;
};
}  // namespace

static int upload_ao_tables(isdf_handle h, const int32_t* atm, int natm, const int32_t* bas, int nbas,
                            const double* env, int nenv, const double* Ls, int nimgs, const double* rcut,
                            const double* phT /* (nimgs, 2) or null */, AoTables* out) {
  ARG_CHECK(h, atm && bas && env && Ls && rcut);
  ARG_CHECK(h, natm > 0 && nbas > 0 && nimgs > 0 && natm <= 65535);
  ARG_CHECK(h, (size_t)nimgs * sizeof(int) <= 64 * 1024);
  constexpr int ATM_SLOTS = 6, BAS_SLOTS = 8, PTR_COORD = 1;
  constexpr int ATOM_OF = 0, ANG_OF = 1, NPRIM_OF = 2, NCTR_OF = 3, PTR_EXP = 5, PTR_COEFF = 6;
  std::vector<AtomDev> atoms(natm);
  std::vector<ShellDev> shells(nbas);
  for (int ia = 0; ia < natm; ++ia) {
    const double* r = env + atm[ia * ATM_SLOTS + PTR_COORD];
    atoms[ia] = {r[0], r[1], r[2], 0.0, -1, -1};
  }
  int ao0 = 0;
  for (int ib = 0; ib < nbas; ++ib) {
    const int32_t* b = bas + ib * BAS_SLOTS;
    const int ia = b[ATOM_OF];
    ARG_CHECK(h, ia >= 0 && ia < natm);
    if (b[ANG_OF] > 3)
      return isdf_fail(h, ISDF_ERR_ARG, "shell %d has l=%d; l <= 3 is supported", ib, b[ANG_OF]);
    if (b[NCTR_OF] > NCMAX)
      return isdf_fail(h, ISDF_ERR_ARG, "shell %d has %d contractions; at most %d supported", ib,
                       b[NCTR_OF], NCMAX);
    ARG_CHECK(h, b[PTR_EXP] + b[NPRIM_OF] <= nenv && b[PTR_COEFF] + b[NPRIM_OF] * b[NCTR_OF] <= nenv);
    shells[ib] = {b[ANG_OF], b[NPRIM_OF], b[NCTR_OF], b[PTR_EXP], b[PTR_COEFF], ao0, rcut[ib] * rcut[ib]};
    ao0 += (2 * b[ANG_OF] + 1) * b[NCTR_OF];
    AtomDev& a = atoms[ia];
    if (a.sh0 < 0) { a.sh0 = ib; a.sh1 = ib + 1; }
    else {
      if (a.sh1 != ib) return isdf_fail(h, ISDF_ERR_ARG, "shells of atom %d are not contiguous in bas", ia);
      a.sh1 = ib + 1;
    }
    if (rcut[ib] > a.rcut_max) a.rcut_max = rcut[ib];
  }
  for (auto& a : atoms) if (a.sh0 < 0) { a.sh0 = a.sh1 = 0; }

  size_t b_atoms = sizeof(AtomDev) * natm, b_shells = sizeof(ShellDev) * nbas;
  size_t b_env = sizeof(double) * nenv, b_Ls = sizeof(double) * 3 * nimgs, b_ph = sizeof(double) * 2 * nimgs;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  char* tab = (char*)isdf_ws(h, "ao_tables", al(b_atoms) + al(b_shells) + al(b_env) + al(b_Ls) + al(b_ph));
  if (!tab) return ISDF_ERR_HIP;
  out->d_atoms = (AtomDev*)tab;
  out->d_shells = (ShellDev*)(tab + al(b_atoms));
  out->d_env = (double*)(tab + al(b_atoms) + al(b_shells));
  out->d_Ls = (double*)(tab + al(b_atoms) + al(b_shells) + al(b_env));
  out->d_phT = (double*)(tab + al(b_atoms) + al(b_shells) + al(b_env) + al(b_Ls));
  out->nao = ao0;
  HIP_TRY(h, hipMemcpyAsync(out->d_atoms, atoms.data(), b_atoms, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(out->d_shells, shells.data(), b_shells, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(out->d_env, env, b_env, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(out->d_Ls, Ls, b_Ls, hipMemcpyHostToDevice, h->stream));
  if (phT) HIP_TRY(h, hipMemcpyAsync(out->d_phT, phT, b_ph, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));   // host staging vectors die at return
  return ISDF_OK;
}

extern "C" int isdf_eval_ao(isdf_handle h, const int32_t* atm, int natm, const int32_t* bas, int nbas,
                            const double* env, int nenv, const double* Ls, int nimgs,
                            const double* rcut, const double* d_coords, int64_t ngrids,
                            double* d_ao, int64_t ld) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_coords && d_ao && ngrids > 0 && ld >= ngrids);
  AoTables t;
  int rc = upload_ao_tables(h, atm, natm, bas, nbas, env, nenv, Ls, nimgs, rcut, nullptr, &t);
  if (rc) return rc;
  dim3 grid((unsigned)cdiv(ngrids, TPB), (unsigned)natm);
  ProfScope ps(h, "eval_ao_kernel[byte]", 8.0 * (double)ngrids * t.nao);
  hipLaunchKernelGGL(eval_ao_kernel, grid, dim3(TPB), (size_t)nimgs * sizeof(int), h->stream,
                     t.d_atoms, t.d_shells, t.d_env, t.d_Ls, nimgs, d_coords, ngrids, d_ao, ld);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_eval_ao_deriv1(isdf_handle h, const int32_t* atm, int natm, const int32_t* bas, int nbas, const double* env,
                                   int nenv, const double* Ls, int nimgs, const double* rcut, const double* d_coords,
                                   int64_t ngrids, double* d_ao, int64_t ld, int64_t plane_stride) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_coords && d_ao && ngrids > 0 && ld >= ngrids);
  AoTables t;
  int rc = upload_ao_tables(h, atm, natm, bas, nbas, env, nenv, Ls, nimgs, rcut, nullptr, &t);
  if (rc) return rc;
  ARG_CHECK(h, plane_stride >= (int64_t)t.nao * ld);
  dim3 grid((unsigned)cdiv(ngrids, TPB), (unsigned)natm);
  ProfScope ps(h, "eval_ao_deriv1_kernel[byte]", 32.0 * (double)ngrids * t.nao);
  hipLaunchKernelGGL(eval_ao_deriv1_kernel, grid, dim3(TPB), (size_t)nimgs * sizeof(int), h->stream, t.d_atoms, t.d_shells,
                     t.d_env, t.d_Ls, nimgs, d_coords, ngrids, d_ao, ld, plane_stride);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_eval_ao_k(isdf_handle h, const int32_t* atm, int natm, const int32_t* bas, int nbas,
                              const double* env, int nenv, const double* Ls, int nimgs,
                              const double* rcut, const double kpt[3], int periodic_part,
                              const double* d_coords, int64_t ngrids, double* d_re, double* d_im,
                              int64_t ld) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, kpt && d_coords && d_re && d_im && ngrids > 0 && ld >= ngrids);
  std::vector<double> ph(2 * (size_t)nimgs);
  for (int i = 0; i < nimgs; ++i) {
    const double kl = kpt[0] * Ls[3 * i] + kpt[1] * Ls[3 * i + 1] + kpt[2] * Ls[3 * i + 2];
    ph[2 * i] = cos(kl);
    ph[2 * i + 1] = sin(kl);
  }
  AoTables t;
  int rc = upload_ao_tables(h, atm, natm, bas, nbas, env, nenv, Ls, nimgs, rcut, ph.data(), &t);
  if (rc) return rc;
  dim3 grid((unsigned)cdiv(ngrids, TPB), (unsigned)natm);
  ProfScope ps(h, "eval_ao_k_kernel[byte]", 16.0 * (double)ngrids * t.nao);
  hipLaunchKernelGGL(eval_ao_k_kernel, grid, dim3(TPB), (size_t)nimgs * sizeof(int), h->stream,
                     t.d_atoms, t.d_shells, t.d_env, t.d_Ls, t.d_phT, nimgs, kpt[0], kpt[1], kpt[2],
                     periodic_part ? 1 : 0, d_coords, ngrids, d_re, d_im, ld);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_eval_ao_k_deriv1(isdf_handle h, const int32_t* atm, int natm, const int32_t* bas, int nbas, const double* env,
                                     int nenv, const double* Ls, int nimgs, const double* rcut, const double kpt[3],
                                     int periodic_part, const double* d_coords, int64_t ngrids, double* d_re, double* d_im,
                                     int64_t ld, int64_t plane_stride) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, kpt && d_coords && d_re && d_im && ngrids > 0 && ld >= ngrids);
  std::vector<double> ph(2 * (size_t)nimgs);
  for (int i = 0; i < nimgs; ++i) {
    const double kl = kpt[0] * Ls[3 * i] + kpt[1] * Ls[3 * i + 1] + kpt[2] * Ls[3 * i + 2];
    ph[2 * i] = cos(kl);
    ph[2 * i + 1] = sin(kl);
  }
  AoTables t;
  int rc = upload_ao_tables(h, atm, natm, bas, nbas, env, nenv, Ls, nimgs, rcut, ph.data(), &t);
  if (rc) return rc;
  ARG_CHECK(h, plane_stride >= (int64_t)t.nao * ld);
  dim3 grid((unsigned)cdiv(ngrids, TPB), (unsigned)natm);
  ProfScope ps(h, "eval_ao_k_deriv1_kernel[byte]", 64.0 * (double)ngrids * t.nao);
  hipLaunchKernelGGL(eval_ao_k_deriv1_kernel, grid, dim3(TPB), (size_t)nimgs * sizeof(int), h->stream, t.d_atoms, t.d_shells,
                     t.d_env, t.d_Ls, t.d_phT, nimgs, kpt[0], kpt[1], kpt[2], periodic_part ? 1 : 0, d_coords, ngrids, d_re, d_im,
                     ld, plane_stride);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_gather_cols(isdf_handle h, const double* d_src, int nrow, int64_t ld_src,
                                const int64_t* d_idx, int64_t n, double* d_dst, int64_t ld_dst) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_src && d_idx && d_dst && nrow > 0 && n > 0 && ld_dst >= n && nrow <= 65535);
  dim3 grid((unsigned)cdiv(n, 256), (unsigned)nrow);
  hipLaunchKernelGGL(gather_cols_kernel, grid, dim3(256), 0, h->stream, d_src, ld_src, d_idx, n, d_dst, ld_dst);
  KERNEL_CHECK(h);
  return ISDF_OK;
}


// out[row * nblk + b] = max |src[row, blk_off[b] .. blk_off[b+1])|: which AO rows vanish identically on which block of grid points
// (the collocation truncates every shell at its rcut, so far AOs are EXACT zeros there and can be left out of a block's selection
// without changing one bit of it)
namespace {
__global__ __launch_bounds__(256) void block_row_absmax_kernel(const double* __restrict__ src, int64_t ld, const int64_t* __restrict__ blk_off,
                                                               int nblk, double* __restrict__ out) {
  __shared__ double red[4];
  const int row = blockIdx.y, b = blockIdx.x;
  const int64_t c0 = blk_off[b], c1 = blk_off[b + 1];
  double m = 0.0;
  for (int64_t c = c0 + threadIdx.x; c < c1; c += 256) m = fmax(m, fabs(src[(int64_t)row * ld + c]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[(int64_t)row * nblk + b] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
}  // namespace

extern "C" int isdf_block_row_absmax(isdf_handle h, const double* d_src, int nrow, int64_t ld, int nblk, const int64_t* d_blk_off,
                                     double* d_out) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_src && d_blk_off && d_out && nrow > 0 && nrow <= 65535 && nblk > 0);
  hipLaunchKernelGGL(block_row_absmax_kernel, dim3((unsigned)nblk, (unsigned)nrow), dim3(256), 0, h->stream, d_src, ld, d_blk_off, nblk,
                     d_out);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

// S4: the Coulomb convolution V = ifft(coulG fft(rows)) of a batch of real fields, hand-written for gfx950.
//
// hipFFT runs a batched 3-D real transform as three unfused passes per direction plus our separate kernel multiply:
// seven kernels, each reading and writing the whole batch (112 G bytes per row against 32 G algorithmic; measured 0.9 TB/s
// algorithmic in round 1).  Here the same arithmetic takes FIVE passes (80 G bytes per row), every one a streaming kernel
// whose 1-D transforms live in LDS:
//
//   1. z, real -> half complex   lines are contiguous; TWO real lines ride one complex transform (a + i b) and are
//                                separated afterwards (A_k = (Z_k + conj Z_{n-k}) / 2, B_k = (Z_k - conj Z_{n-k}) / 2i)
//   2. y forward                 lines strided by n2h; a workgroup takes all n1 elements of 16 ADJACENT lines, so every global
//                                access is a 256-byte run and the LDS layout [element][line] is bank-conflict free
//   3. x forward, x coulG, x inverse   the same kernel along the slowest axis, kernel multiply fused between the two
//                                transforms (the table already carries the 1/G of the inverse, pbc.py:182-211)
//   4. y inverse
//   5. z, half complex -> real   the inverse of pass 1 (imaginary parts of the self-conjugate entries are ignored, as
//                                hipFFT's Z2D does)
//
// The 1-D transform is a Stockham autosort FFT (decimation in frequency, out of place between two LDS buffers, natural order
// in and out) over the mixed radices of n: 4, 2, 3, 5 hard-coded, 7 / 11 / 13 by a generic O(r^2) butterfly; twiddles
// W_n^k = exp(-2 pi i k / n) come from a host-made double table.  One butterfly per thread and stage, consecutive threads on
// consecutive lines.  Meshes with a prime factor above 13 (or a dimension above 1024) fall back to hipFFT in coulomb.hip.
//
// Two implementations of the stages.  FAST (every dimension 2-3-5 smooth: all production meshes): ONE LDS buffer - a stage
// reads its butterflies' inputs, keeps the outputs in registers across a barrier and writes them back in place, so a
// workgroup needs n x 16 lines x 16 B <= 32 KB and five workgroups share a CU (the passes are latency chains of load /
// stages / store: occupancy is what hides them); twiddles staged in LDS; the number of lines per tile is a compile-time
// power of two and the divisions by the stage stride go through a float reciprocal.  GENERIC (a factor 7, 11 or 13
// somewhere): two LDS buffers, plain Stockham ping-pong.
//
// Reference semantics: pyscf/pbc/tools/pbc.py:149-211 (fft unscaled, ifft 1/N, C order over the mesh).
#include "common.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int TPB = 256;
constexpr int MAXSTAGE = 12;

struct Axis {
  int n;
  int nstage;
  int radix[MAXSTAGE];
  const double2* tw;     // n entries, exp(-2 pi i k / n)
};

__device__ inline double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ inline double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ inline double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// multiply by SIGN * i
template <int SIGN>
__device__ inline double2 mul_si(double2 a) { return SIGN > 0 ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x); }
template <int SIGN>
__device__ inline double2 twid(const double2* __restrict__ tw, int idx) {
  double2 w = tw[idx];
  if (SIGN > 0) w.y = -w.y;
  return w;
}

// One Stockham stage of radix r on M lines held as x[element * Ls + line]:  n_cur = current sub-length, s = stride.
template <int SIGN>
__device__ inline void stockham_stage(const double2* __restrict__ x, double2* __restrict__ y, int N, int n_cur, int s, int r,
                                      int Ls, int M, const double2* __restrict__ tw) {
  const int mm = n_cur / r;
  const int nbf = N / r;
  for (int i = threadIdx.x; i < nbf * M; i += TPB) {
    const int m = i % M, bf = i / M;
    const int q = bf % s, p = bf / s;
    const double2* xi = x + (int64_t)(q + s * p) * Ls + m;
    double2* yo = y + (int64_t)(q + s * r * p) * Ls + m;
    const int sin_ = s * mm * Ls;      // input step between the r legs
    const int sout = s * Ls;           // output step
    const int tstep = (p * s) % N;     // twiddle exponent of leg 1
    if (r == 4) {
      const double2 a0 = xi[0], a1 = xi[sin_], a2 = xi[2 * sin_], a3 = xi[3 * sin_];
      const double2 u0 = cadd(a0, a2), u1 = csub(a0, a2), u2 = cadd(a1, a3), u3 = mul_si<SIGN>(csub(a1, a3));
      yo[0] = cadd(u0, u2);
      yo[sout] = cmul(cadd(u1, u3), twid<SIGN>(tw, tstep));
      yo[2 * sout] = cmul(csub(u0, u2), twid<SIGN>(tw, (2 * tstep) % N));
      yo[3 * sout] = cmul(csub(u1, u3), twid<SIGN>(tw, (3 * tstep) % N));
    } else if (r == 2) {
      const double2 a0 = xi[0], a1 = xi[sin_];
      yo[0] = cadd(a0, a1);
      yo[sout] = cmul(csub(a0, a1), twid<SIGN>(tw, tstep));
    } else if (r == 3) {
      const double2 a0 = xi[0], a1 = xi[sin_], a2 = xi[2 * sin_];
      const double2 t1 = cadd(a1, a2);
      const double2 t2 = make_double2(a0.x - 0.5 * t1.x, a0.y - 0.5 * t1.y);
      const double2 d = csub(a1, a2);
      const double2 t3 = mul_si<SIGN>(make_double2(0.86602540378443864676 * d.x, 0.86602540378443864676 * d.y));
      yo[0] = cadd(a0, t1);
      yo[sout] = cmul(cadd(t2, t3), twid<SIGN>(tw, tstep));
      yo[2 * sout] = cmul(csub(t2, t3), twid<SIGN>(tw, (2 * tstep) % N));
    } else if (r == 5) {
      const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
      const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
      const double2 a0 = xi[0], a1 = xi[sin_], a2 = xi[2 * sin_], a3 = xi[3 * sin_], a4 = xi[4 * sin_];
      const double2 t1 = cadd(a1, a4), t2 = cadd(a2, a3), t3 = csub(a1, a4), t4 = csub(a2, a3);
      const double2 r1 = make_double2(a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y);
      const double2 r2 = make_double2(a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y);
      const double2 i1 = mul_si<SIGN>(make_double2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
      const double2 i2 = mul_si<SIGN>(make_double2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
      yo[0] = make_double2(a0.x + t1.x + t2.x, a0.y + t1.y + t2.y);
      yo[sout] = cmul(cadd(r1, i1), twid<SIGN>(tw, tstep));
      yo[2 * sout] = cmul(cadd(r2, i2), twid<SIGN>(tw, (2 * tstep) % N));
      yo[3 * sout] = cmul(csub(r2, i2), twid<SIGN>(tw, (3 * tstep) % N));
      yo[4 * sout] = cmul(csub(r1, i1), twid<SIGN>(tw, (4 * tstep) % N));
    } else {
      // generic prime radix: b_j = sum_k a_k w_r^{jk}, w_r^t = W_N^{(N/r) t}
      const int wr = N / r;
      for (int j = 0; j < r; ++j) {
        double2 b = xi[0];
        for (int k = 1; k < r; ++k) b = cadd(b, cmul(xi[k * sin_], twid<SIGN>(tw, wr * ((j * k) % r))));
        yo[j * sout] = cmul(b, twid<SIGN>(tw, (int)(((int64_t)j * tstep) % N)));
      }
    }
  }
}

// All stages; returns the buffer that holds the result (natural order).
template <int SIGN>
__device__ inline double2* stockham_all(double2* x, double2* y, const Axis& ax, int Ls, int M) {
  int n_cur = ax.n, s = 1;
  for (int st = 0; st < ax.nstage; ++st) {
    const int r = ax.radix[st];
    stockham_stage<SIGN>(x, y, ax.n, n_cur, s, r, Ls, M, ax.tw);
    __syncthreads();
    n_cur /= r;
    s *= r;
    double2* t = x; x = y; y = t;
  }
  return x;
}

// Pass 1: real lines (n contiguous doubles each, nlines of them back to back) -> half-complex lines (nh = n/2 + 1).
// A workgroup takes LP line pairs.
__global__ __launch_bounds__(TPB) void z_r2c_kernel(const double* __restrict__ in, double2* __restrict__ out, int64_t nlines,
                                                    Axis ax, int LP) {
  extern __shared__ double2 lds[];
  const int n = ax.n, nh = n / 2 + 1, Ls = LP + 1;
  double2* x = lds;
  double2* y = lds + (int64_t)n * Ls;
  const int64_t line0 = (int64_t)blockIdx.x * 2 * LP;
  const int nl = (int)min((int64_t)2 * LP, nlines - line0);     // lines of this tile
  const int M = (nl + 1) / 2;
  const double* src = in + line0 * n;
  for (int c = threadIdx.x; c < 2 * M * n; c += TPB) {
    const int l = c / n, e = c - l * n;
    const double v = (l < nl) ? src[c] : 0.0;
    double* dst = (double*)(x + (int64_t)e * Ls + (l >> 1));
    dst[l & 1] = v;
  }
  __syncthreads();
  const double2* z = stockham_all<-1>(x, y, ax, Ls, M);
  double2* dstc = out + line0 * nh;
  for (int c = threadIdx.x; c < nl * nh; c += TPB) {
    const int l = c / nh, k = c - l * nh;
    const int m = l >> 1;
    const double2 zk = z[(int64_t)k * Ls + m];
    const double2 zn = z[(int64_t)((n - k) % n) * Ls + m];
    double2 r;
    if ((l & 1) == 0) r = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));        // (Z_k + conj Z_{n-k}) / 2
    else r = make_double2(0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x));                   // (Z_k - conj Z_{n-k}) / 2i
    dstc[c] = r;
  }
}

// Pass 5: half-complex lines -> real lines (unnormalised inverse; the 1/N sits in the kernel table).
__global__ __launch_bounds__(TPB) void z_c2r_kernel(const double2* __restrict__ in, double* __restrict__ out, int64_t nlines,
                                                    Axis ax, int LP) {
  extern __shared__ double2 lds[];
  const int n = ax.n, nh = n / 2 + 1, Ls = LP + 1;
  double2* x = lds;
  double2* y = lds + (int64_t)n * Ls;
  const int64_t line0 = (int64_t)blockIdx.x * 2 * LP;
  const int nl = (int)min((int64_t)2 * LP, nlines - line0);
  const int M = (nl + 1) / 2;
  const double2* src = in + line0 * nh;
  // Z_k = A_k + i B_k (k <= n/2),  Z_{n-k} = conj(A_k) + i conj(B_k);  A = even line, B = odd line of the pair
  for (int c = threadIdx.x; c < M * nh; c += TPB) {
    const int m = c / nh, k = c - m * nh;
    double2 a = src[(int64_t)(2 * m) * nh + k];
    double2 b = (2 * m + 1 < nl) ? src[(int64_t)(2 * m + 1) * nh + k] : make_double2(0.0, 0.0);
    const bool selfconj = (k == 0) || (2 * k == n);
    if (selfconj) { a.y = 0.0; b.y = 0.0; }
    x[(int64_t)k * Ls + m] = make_double2(a.x - b.y, a.y + b.x);
    if (!selfconj) x[(int64_t)(n - k) * Ls + m] = make_double2(a.x + b.y, -a.y + b.x);
  }
  __syncthreads();
  const double2* z = stockham_all<1>(x, y, ax, Ls, M);
  double* dst = out + line0 * n;
  for (int c = threadIdx.x; c < nl * n; c += TPB) {
    const int l = c / n, e = c - l * n;
    const double* zz = (const double*)(z + (int64_t)e * Ls + (l >> 1));
    dst[c] = zz[l & 1];
  }
}

// Passes 2-4: 1-D transforms along a strided axis, in place.  data[o * os + e * es + m], e < n (the transformed axis),
// m < Mtot adjacent lines (unit stride), o outer blocks.  MODE 0: forward, 1: inverse, 2: forward, multiply by
// table[e * es + m], inverse.
template <int MODE>
__global__ __launch_bounds__(TPB) void strided_fft_kernel(double2* __restrict__ data, int64_t os, int64_t es, int Mtot, Axis ax,
                                                          int ZC, int ntile, const double* __restrict__ table) {
  extern __shared__ double2 lds[];
  const int n = ax.n;
  double2* x = lds;
  double2* y = lds + (int64_t)n * ZC;
  const int64_t outer = blockIdx.x / ntile;              // 1-D grid: (outer block, tile of adjacent lines)
  const int m0 = (int)(blockIdx.x % ntile) * ZC;
  const int M = min(ZC, Mtot - m0);
  double2* base = data + outer * os + m0;
  for (int c = threadIdx.x; c < n * ZC; c += TPB) {
    const int e = c / ZC, m = c - e * ZC;
    if (m < M) x[(int64_t)e * ZC + m] = base[(int64_t)e * es + m];
  }
  __syncthreads();
  double2* z;
  if (MODE == 1) {
    z = stockham_all<1>(x, y, ax, ZC, M);
  } else {
    z = stockham_all<-1>(x, y, ax, ZC, M);
    if (MODE == 2) {
      const double* tb = table + m0;
      for (int c = threadIdx.x; c < n * ZC; c += TPB) {
        const int e = c / ZC, m = c - e * ZC;
        if (m < M) {
          const double g = tb[(int64_t)e * es + m];
          double2 v = z[(int64_t)e * ZC + m];
          v.x *= g;
          v.y *= g;
          z[(int64_t)e * ZC + m] = v;
        }
      }
      __syncthreads();
      double2* other = (z == x) ? y : x;
      z = stockham_all<1>(z, other, ax, ZC, M);
    }
  }
  for (int c = threadIdx.x; c < n * ZC; c += TPB) {
    const int e = c / ZC, m = c - e * ZC;
    if (m < M) base[(int64_t)e * es + m] = z[(int64_t)e * ZC + m];
  }
}


// ---- FAST path: in-place stages with register-held outputs -----------------------------------------------------------
// Invariant (host): n * ZCT <= 2048, so a stage has at most 8 / R butterflies per thread.
template <int SIGN, int R>
__device__ inline void butterfly_r(const double2* a, double2* b) {
  if (R == 2) {
    b[0] = cadd(a[0], a[1]);
    b[1] = csub(a[0], a[1]);
  } else if (R == 3) {
    const double2 t1 = cadd(a[1], a[2]);
    const double2 t2 = make_double2(a[0].x - 0.5 * t1.x, a[0].y - 0.5 * t1.y);
    const double2 d = csub(a[1], a[2]);
    const double2 t3 = mul_si<SIGN>(make_double2(0.86602540378443864676 * d.x, 0.86602540378443864676 * d.y));
    b[0] = cadd(a[0], t1);
    b[1] = cadd(t2, t3);
    b[2] = csub(t2, t3);
  } else if (R == 4) {
    const double2 u0 = cadd(a[0], a[2]), u1 = csub(a[0], a[2]), u2 = cadd(a[1], a[3]), u3 = mul_si<SIGN>(csub(a[1], a[3]));
    b[0] = cadd(u0, u2);
    b[1] = cadd(u1, u3);
    b[2] = csub(u0, u2);
    b[3] = csub(u1, u3);
  } else {
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
    const double2 t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
    const double2 r1 = make_double2(a[0].x + c1 * t1.x + c2 * t2.x, a[0].y + c1 * t1.y + c2 * t2.y);
    const double2 r2 = make_double2(a[0].x + c2 * t1.x + c1 * t2.x, a[0].y + c2 * t1.y + c1 * t2.y);
    const double2 i1 = mul_si<SIGN>(make_double2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
    const double2 i2 = mul_si<SIGN>(make_double2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
    b[0] = make_double2(a[0].x + t1.x + t2.x, a[0].y + t1.y + t2.y);
    b[1] = cadd(r1, i1);
    b[2] = cadd(r2, i2);
    b[3] = csub(r2, i2);
    b[4] = csub(r1, i1);
  }
}

// exp(-2 pi i k / R) for the composite radices, as literals: the constant twiddles inside a composite butterfly then cost no
// LDS read (they were 6 of the 29 reads of a radix-12 butterfly) and the trivial ones (1, -i, ...) fold away.  k is a compile-time
// constant wherever this is called (fully unrolled loops).
template <int SIGN, int R>
__device__ inline double2 const_twid(int k) {
  if constexpr (R == 6) {
    constexpr double c[6] = {1.0, 0.5000000000000001, -0.4999999999999998, -1.0, -0.5000000000000004, 0.5000000000000001};
    constexpr double sn[6] = {-0.0, -0.8660254037844386, -0.8660254037844387, -1.2246467991473532e-16, 0.8660254037844384, 0.8660254037844386};
    return make_double2(c[k], SIGN > 0 ? -sn[k] : sn[k]);
  }
  else if constexpr (R == 8) {
    constexpr double c[8] = {1.0, 0.7071067811865476, 6.123233995736766e-17, -0.7071067811865475, -1.0, -0.7071067811865477, -1.8369701987210297e-16, 0.7071067811865474};
    constexpr double sn[8] = {-0.0, -0.7071067811865475, -1.0, -0.7071067811865476, -1.2246467991473532e-16, 0.7071067811865475, 1.0, 0.7071067811865477};
    return make_double2(c[k], SIGN > 0 ? -sn[k] : sn[k]);
  }
  else if constexpr (R == 9) {
    constexpr double c[9] = {1.0, 0.766044443118978, 0.17364817766693041, -0.4999999999999998, -0.9396926207859083, -0.9396926207859084, -0.5000000000000004, 0.17364817766692997, 0.7660444431189778};
    constexpr double sn[9] = {-0.0, -0.6427876096865393, -0.984807753012208, -0.8660254037844387, -0.3420201433256689, 0.34202014332566866, 0.8660254037844384, 0.9848077530122081, 0.6427876096865396};
    return make_double2(c[k], SIGN > 0 ? -sn[k] : sn[k]);
  }
  else if constexpr (R == 10) {
    constexpr double c[10] = {1.0, 0.8090169943749475, 0.30901699437494745, -0.30901699437494734, -0.8090169943749473, -1.0, -0.8090169943749476, -0.30901699437494756, 0.30901699437494723, 0.8090169943749473};
    constexpr double sn[10] = {-0.0, -0.5877852522924731, -0.9510565162951535, -0.9510565162951536, -0.5877852522924732, -1.2246467991473532e-16, 0.587785252292473, 0.9510565162951535, 0.9510565162951536, 0.5877852522924734};
    return make_double2(c[k], SIGN > 0 ? -sn[k] : sn[k]);
  }
  else if constexpr (R == 12) {
    constexpr double c[12] = {1.0, 0.8660254037844387, 0.5000000000000001, 6.123233995736766e-17, -0.4999999999999998, -0.8660254037844387, -1.0, -0.8660254037844388, -0.5000000000000004, -1.8369701987210297e-16, 0.5000000000000001, 0.8660254037844384};
    constexpr double sn[12] = {-0.0, -0.49999999999999994, -0.8660254037844386, -1.0, -0.8660254037844387, -0.49999999999999994, -1.2246467991473532e-16, 0.4999999999999997, 0.8660254037844384, 1.0, 0.8660254037844386, 0.5000000000000004};
    return make_double2(c[k], SIGN > 0 ? -sn[k] : sn[k]);
  }
  else if constexpr (R == 15) {
    constexpr double c[15] = {1.0, 0.9135454576426009, 0.6691306063588582, 0.30901699437494745, -0.10452846326765333, -0.4999999999999998, -0.8090169943749473, -0.9781476007338057, -0.9781476007338057, -0.8090169943749476, -0.5000000000000004, -0.10452846326765423, 0.30901699437494723, 0.6691306063588585, 0.913545457642601};
    constexpr double sn[15] = {-0.0, -0.40673664307580015, -0.7431448254773941, -0.9510565162951535, -0.9945218953682734, -0.8660254037844387, -0.5877852522924732, -0.20791169081775931, 0.20791169081775907, 0.587785252292473, 0.8660254037844384, 0.9945218953682733, 0.9510565162951536, 0.743144825477394, 0.40673664307580015};
    return make_double2(c[k], SIGN > 0 ? -sn[k] : sn[k]);
  }
  else if constexpr (R == 16) {
    constexpr double c[16] = {1.0, 0.9238795325112867, 0.7071067811865476, 0.38268343236508984, 6.123233995736766e-17, -0.3826834323650897, -0.7071067811865475, -0.9238795325112867, -1.0, -0.9238795325112868, -0.7071067811865477, -0.38268343236509034, -1.8369701987210297e-16, 0.38268343236509, 0.7071067811865474, 0.9238795325112865};
    constexpr double sn[16] = {-0.0, -0.3826834323650898, -0.7071067811865475, -0.9238795325112867, -1.0, -0.9238795325112867, -0.7071067811865476, -0.3826834323650899, -1.2246467991473532e-16, 0.38268343236508967, 0.7071067811865475, 0.9238795325112865, 1.0, 0.9238795325112866, 0.7071067811865477, 0.3826834323650904};
    return make_double2(c[k], SIGN > 0 ? -sn[k] : sn[k]);
  }
  else return make_double2(1.0, 0.0);
}

// Composite radices in registers: R = R1 R2 point DFT as R2 DFTs of length R1 (inputs R2 apart), the constant twiddles
// W_R^(n2 k1) = tw[n2 k1 N / R] (R divides N, so the axis table holds them), then R1 DFTs of length R2; output index k1 + R1 k2.
// A 120-point line is then two stages (15 x 8) instead of four (4 x 2 x 3 x 5): half the LDS round trips and barriers, which is
// what bounds these passes.
template <int SIGN, int R1, int R2, bool CT>
__device__ inline void butterfly_comp(const double2* a, double2* b, const double2* __restrict__ tw, int N) {
  constexpr int R = R1 * R2;
  double2 t[R2][R1];
#pragma unroll
  for (int n2 = 0; n2 < R2; ++n2) {
    double2 u[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) u[n1] = a[R2 * n1 + n2];
    butterfly_r<SIGN, R1>(u, t[n2]);
  }
  // CT: literals (no LDS read; needs a few more registers for the constants - the 1024-thread plane kernels, at 113-115 of their
  // 128 VGPRs, keep the table)
  const int step = N / R;
#pragma unroll
  for (int n2 = 1; n2 < R2; ++n2)
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1)
      t[n2][k1] = cmul(t[n2][k1], CT ? const_twid<SIGN, R>(n2 * k1) : twid<SIGN>(tw, n2 * k1 * step));
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) {
    double2 v[R2], w[R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) v[n2] = t[n2][k1];
    butterfly_r<SIGN, R2>(v, w);
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) b[k1 + R1 * k2] = w[k2];
  }
}

template <int SIGN, int R, bool CT = true>
__device__ inline void butterfly_any(const double2* a, double2* b, const double2* __restrict__ tw, int N) {
  if (R <= 5) butterfly_r<SIGN, (R <= 5 ? R : 2)>(a, b);
  else if (R == 6) butterfly_comp<SIGN, 3, 2, CT>(a, b, tw, N);
  else if (R == 8) butterfly_comp<SIGN, 4, 2, CT>(a, b, tw, N);
  else if (R == 9) butterfly_comp<SIGN, 3, 3, CT>(a, b, tw, N);
  else if (R == 10) butterfly_comp<SIGN, 5, 2, CT>(a, b, tw, N);
  else if (R == 12) butterfly_comp<SIGN, 4, 3, CT>(a, b, tw, N);
  else if (R == 15) butterfly_comp<SIGN, 5, 3, CT>(a, b, tw, N);
  else butterfly_comp<SIGN, 4, 4, CT>(a, b, tw, N);        // 16
}

template <int SIGN, int R, int ZCT>
__device__ inline void stage_inplace(double2* __restrict__ buf, int N, int n_cur, int s, int Ls,
                                     const double2* __restrict__ tw) {
  constexpr int KMAX = (8 + R - 1) / R;
  const int mm = n_cur / R;
  const int total = (N / R) * ZCT;
  const float inv_s = 1.0f / (float)s;
  double2 out[KMAX][R];
#pragma unroll
  for (int u = 0; u < KMAX; ++u) {
    const int i = threadIdx.x + u * TPB;
    if (i < total) {
      const int m = i & (ZCT - 1), bf = i / ZCT;
      const int p = (int)(((float)bf + 0.5f) * inv_s), q = bf - p * s;
      const double2* xi = buf + (q + s * p) * Ls + m;
      const int sin_ = s * mm * Ls;
      double2 a[R];
#pragma unroll
      for (int k = 0; k < R; ++k) a[k] = xi[k * sin_];
      butterfly_any<SIGN, R>(a, out[u], tw, N);
      if (mm > 1) {                      // (the last stage has p = 0: every output twiddle is 1)
        const int tstep = p * s;         // < N
        int t = tstep;
#pragma unroll
        for (int j = 1; j < R; ++j) {
          out[u][j] = cmul(out[u][j], twid<SIGN>(tw, t));
          t += tstep;
          if (t >= N) t -= N;
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < KMAX; ++u) {
    const int i = threadIdx.x + u * TPB;
    if (i < total) {
      const int m = i & (ZCT - 1), bf = i / ZCT;
      const int p = (int)(((float)bf + 0.5f) * inv_s), q = bf - p * s;
      double2* yo = buf + (q + s * R * p) * Ls + m;
      const int sout = s * Ls;
#pragma unroll
      for (int j = 0; j < R; ++j) yo[j * sout] = out[u][j];
    }
  }
  __syncthreads();
}

template <int SIGN, int ZCT>
__device__ inline void fft_inplace(double2* buf, const Axis& ax, int Ls, const double2* tw) {
  int n_cur = ax.n, s = 1;
  for (int st = 0; st < ax.nstage; ++st) {
    const int r = ax.radix[st];
    switch (r) {
      case 16: stage_inplace<SIGN, 16, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
      case 15: stage_inplace<SIGN, 15, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
      case 12: stage_inplace<SIGN, 12, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
      case 10: stage_inplace<SIGN, 10, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
      case 9: stage_inplace<SIGN, 9, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
      case 8: stage_inplace<SIGN, 8, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
      case 6: stage_inplace<SIGN, 6, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
      case 5: stage_inplace<SIGN, 5, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
      case 4: stage_inplace<SIGN, 4, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
      case 3: stage_inplace<SIGN, 3, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
      default: stage_inplace<SIGN, 2, ZCT>(buf, ax.n, n_cur, s, Ls, tw); break;
    }
    n_cur /= r;
    s *= r;
  }
}

// LDS: [n * Ls complex work buffer][n complex twiddles]
template <int ZCT>
__global__ __launch_bounds__(TPB) void z_r2c_fast_kernel(const double* __restrict__ in, double2* __restrict__ out,
                                                         int64_t nlines, Axis ax) {
  extern __shared__ double2 lds[];
  const int n = ax.n, nh = n / 2 + 1;
  constexpr int Ls = ZCT + 1;
  double2* x = lds;
  double2* tw = lds + n * Ls;
  for (int k = threadIdx.x; k < n; k += TPB) tw[k] = ax.tw[k];
  const int64_t line0 = (int64_t)blockIdx.x * 2 * ZCT;
  const int nl = (int)min((int64_t)2 * ZCT, nlines - line0);
  const double* src = in + line0 * n;
  const float inv_n = 1.0f / (float)n;
  for (int c = threadIdx.x; c < 2 * ZCT * n; c += TPB) {
    const int l = (int)(((float)c + 0.5f) * inv_n), e = c - l * n;
    const double v = (l < nl) ? src[c] : 0.0;
    ((double*)(x + e * Ls + (l >> 1)))[l & 1] = v;
  }
  __syncthreads();
  fft_inplace<-1, ZCT>(x, ax, Ls, tw);
  double2* dstc = out + line0 * nh;
  const float inv_nh = 1.0f / (float)nh;
  for (int c = threadIdx.x; c < nl * nh; c += TPB) {
    const int l = (int)(((float)c + 0.5f) * inv_nh), k = c - l * nh;
    const int m = l >> 1;
    const double2 zk = x[k * Ls + m];
    const double2 zn = x[(k == 0 ? 0 : n - k) * Ls + m];
    double2 r;
    if ((l & 1) == 0) r = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
    else r = make_double2(0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x));
    dstc[c] = r;
  }
}

template <int ZCT>
__global__ __launch_bounds__(TPB) void z_c2r_fast_kernel(const double2* __restrict__ in, double* __restrict__ out,
                                                         int64_t nlines, Axis ax) {
  extern __shared__ double2 lds[];
  const int n = ax.n, nh = n / 2 + 1;
  constexpr int Ls = ZCT + 1;
  double2* x = lds;
  double2* tw = lds + n * Ls;
  for (int k = threadIdx.x; k < n; k += TPB) tw[k] = ax.tw[k];
  const int64_t line0 = (int64_t)blockIdx.x * 2 * ZCT;
  const int nl = (int)min((int64_t)2 * ZCT, nlines - line0);
  const int M = (nl + 1) / 2;
  const double2* src = in + line0 * nh;
  const float inv_nh = 1.0f / (float)nh;
  // Z_k = A_k + i B_k (k <= n/2),  Z_{n-k} = conj(A_k) + i conj(B_k);  A = even line, B = odd line of the pair
  for (int c = threadIdx.x; c < M * nh; c += TPB) {
    const int m = (int)(((float)c + 0.5f) * inv_nh), k = c - m * nh;
    double2 a = src[(int64_t)(2 * m) * nh + k];
    double2 b = (2 * m + 1 < nl) ? src[(int64_t)(2 * m + 1) * nh + k] : make_double2(0.0, 0.0);
    const bool selfconj = (k == 0) || (2 * k == n);
    if (selfconj) { a.y = 0.0; b.y = 0.0; }
    x[k * Ls + m] = make_double2(a.x - b.y, a.y + b.x);
    if (!selfconj) x[(n - k) * Ls + m] = make_double2(a.x + b.y, -a.y + b.x);
  }
  __syncthreads();
  fft_inplace<1, ZCT>(x, ax, Ls, tw);
  double* dst = out + line0 * n;
  const float inv_n = 1.0f / (float)n;
  for (int c = threadIdx.x; c < nl * n; c += TPB) {
    const int l = (int)(((float)c + 0.5f) * inv_n), e = c - l * n;
    dst[c] = ((const double*)(x + e * Ls + (l >> 1)))[l & 1];
  }
}

template <int MODE, int ZCT>
__global__ __launch_bounds__(TPB) void strided_fft_fast_kernel(double2* __restrict__ data, int64_t os, int64_t es, int Mtot,
                                                               Axis ax, int ntile, const double* __restrict__ table) {
  extern __shared__ double2 lds[];
  const int n = ax.n;
  double2* x = lds;
  double2* tw = lds + n * ZCT;
  for (int k = threadIdx.x; k < n; k += TPB) tw[k] = ax.tw[k];
  // MODE 2 walks tile-major (consecutive workgroups = the same tile of consecutive rows): the kernel-table tile they all
  // multiply with stays in L2 instead of being fetched once per row (3.6 GB of 18.7 GB per 512-row launch in the PMC
  // counters); the plain transforms walk row-major
  const int nouter = gridDim.x / ntile;
  const int64_t outer = (MODE == 2) ? blockIdx.x % nouter : blockIdx.x / ntile;
  const int m0 = (int)((MODE == 2) ? blockIdx.x / nouter : blockIdx.x % ntile) * ZCT;
  const int M = min(ZCT, Mtot - m0);
  double2* base = data + outer * os + m0;
  for (int c = threadIdx.x; c < n * ZCT; c += TPB) {
    const int e = c / ZCT, m = c & (ZCT - 1);
    if (m < M) x[c] = base[(int64_t)e * es + m];
  }
  __syncthreads();
  if (MODE == 1) {
    fft_inplace<1, ZCT>(x, ax, ZCT, tw);
  } else {
    fft_inplace<-1, ZCT>(x, ax, ZCT, tw);
    if (MODE == 2) {
      const double* tb = table + m0;
      for (int c = threadIdx.x; c < n * ZCT; c += TPB) {
        const int e = c / ZCT, m = c & (ZCT - 1);
        if (m < M) {
          const double g = tb[(int64_t)e * es + m];
          double2 v = x[c];
          v.x *= g;
          v.y *= g;
          x[c] = v;
        }
      }
      __syncthreads();
      fft_inplace<1, ZCT>(x, ax, ZCT, tw);
    }
  }
  for (int c = threadIdx.x; c < n * ZCT; c += TPB) {
    const int e = c / ZCT, m = c & (ZCT - 1);
    if (m < M) base[(int64_t)e * es + m] = x[c];
  }
}

// ---- PLANE path: z and y transforms of one (y, z) plane fused in LDS: three passes instead of five ---------------------------
// A workgroup of 1024 threads owns the plane of one (row, x): n1 real lines of n2 points = 115 KB at 120^3, the half-complex plane
// n1 x (n2/2+1) = 117 KB - the whole of it stays in LDS between the z and the y transform, so the pair
//   forward:  real plane -> z real-to-half-complex (two lines per complex transform) -> y forward -> complex plane
//   inverse:  complex plane -> y inverse -> z half-complex-to-real -> real plane
// reads and writes each plane ONCE (48 G bytes per row for the whole convolution with the fused x pass in between, against 80 G
// for the five passes; algorithmic 32 G).  One workgroup per CU; stages as in the FAST path (in place, outputs held in
// registers across a barrier), generalised to any number of lines and 1024 threads.  Layouts: z stage x[z][pair] with the pitch
// chosen = 1 mod 16 (the real-plane load and the separation walk z fastest: a stride of 4 banks), y stage P[y][kz] with pitch
// n2/2+1 = the global layout of the half spectrum.
constexpr int TPBP = 1024;
constexpr int PLANE_ENTRIES = 10240;
constexpr int PIPE_NT = 768;                     // threads of the pipelined plane passes (12 waves: 170 VGPRs each)             // complex numbers: the most a plane buffer may hold (160 KB of LDS)

template <int SIGN, int R, int NT, int KM = 0>
__device__ inline void stage_plane(double2* __restrict__ buf, int N, int n_cur, int s, int Ls, int M,
                                   const double2* __restrict__ tw, const int tid) {
  constexpr int KMAX = KM > 0 ? KM : ((PLANE_ENTRIES + NT - 1) / NT + R - 1) / R;
  const int mm = n_cur / R;
  const int total = (N / R) * M;
  const float inv_s = 1.0f / (float)s, inv_M = 1.0f / (float)M;
  double2 out[KMAX][R];
#pragma unroll
  for (int u = 0; u < KMAX; ++u) {
    const int i = tid + u * NT;
    if (i < total) {
      const int bf = (int)(((float)i + 0.5f) * inv_M), m = i - bf * M;
      const int p = (int)(((float)bf + 0.5f) * inv_s), q = bf - p * s;
      const double2* xi = buf + (q + s * p) * Ls + m;
      const int sin_ = s * mm * Ls;
      double2 a[R];
#pragma unroll
      for (int k = 0; k < R; ++k) a[k] = xi[k * sin_];
      butterfly_any<SIGN, R, (KM > 0)>(a, out[u], tw, N);
      if (KM == 0 || mm > 1) {           // (the last stage has p = 0: every output twiddle is 1; skipped in the sized kernels)
        const int tstep = p * s;
        int t = tstep;
#pragma unroll
        for (int j = 1; j < R; ++j) {
          out[u][j] = cmul(out[u][j], twid<SIGN>(tw, t));
          t += tstep;
          if (t >= N) t -= N;
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < KMAX; ++u) {
    const int i = tid + u * NT;
    if (i < total) {
      const int bf = (int)(((float)i + 0.5f) * inv_M), m = i - bf * M;
      const int p = (int)(((float)bf + 0.5f) * inv_s), q = bf - p * s;
      double2* yo = buf + (q + s * R * p) * Ls + m;
      const int sout = s * Ls;
#pragma unroll
      for (int j = 0; j < R; ++j) yo[j * sout] = out[u][j];
    }
  }
  __syncthreads();
}

template <int SIGN, int NT = 1024>
__device__ inline void fft_plane(double2* buf, const Axis& ax, int Ls, int M, const double2* tw) {
  int n_cur = ax.n, s = 1;
  for (int st = 0; st < ax.nstage; ++st) {
    const int r = ax.radix[st];
    switch (r) {
      case 16: stage_plane<SIGN, 16, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
      case 15: stage_plane<SIGN, 15, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
      case 12: stage_plane<SIGN, 12, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
      case 10: stage_plane<SIGN, 10, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
      case 9: stage_plane<SIGN, 9, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
      case 8: stage_plane<SIGN, 8, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
      case 6: stage_plane<SIGN, 6, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
      case 5: stage_plane<SIGN, 5, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
      case 4: stage_plane<SIGN, 4, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
      case 3: stage_plane<SIGN, 3, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
      default: stage_plane<SIGN, 2, NT>(buf, ax.n, n_cur, s, Ls, M, tw, (int)threadIdx.x); break;
    }
    n_cur /= r;
    s *= r;
  }
}

// forward: in = real planes (n1 x n2 each, back to back), out = half-complex planes (n1 x n2h), y already transformed
__global__ __launch_bounds__(TPBP) void plane_fwd_kernel(const double* __restrict__ in, double2* __restrict__ out, Axis az, Axis ay,
                                                         int Lz, int bufsz) {
  extern __shared__ double2 lds[];
  const int n2 = az.n, n1 = ay.n, n2h = n2 / 2 + 1, npair = (n1 + 1) / 2;
  double2* x = lds;
  double2* twz = lds + bufsz;
  double2* twy = twz + n2;
  for (int k = threadIdx.x; k < n2; k += TPBP) twz[k] = az.tw[k];
  for (int k = threadIdx.x; k < n1; k += TPBP) twy[k] = ay.tw[k];
  const double* src = in + (int64_t)blockIdx.x * n1 * n2;
  double2* dst = out + (int64_t)blockIdx.x * n1 * n2h;
  if (n1 & 1)                                           // the last line has no partner: its imaginary slot is zero
    for (int e = threadIdx.x; e < n2; e += TPBP) x[e * Lz + npair - 1].y = 0.0;
  const float inv_n2 = 1.0f / (float)n2, inv_nh = 1.0f / (float)n2h;
  for (int c = threadIdx.x; c < n1 * n2; c += TPBP) {
    const int l = (int)(((float)c + 0.5f) * inv_n2), e = c - l * n2;
    ((double*)(x + e * Lz + (l >> 1)))[l & 1] = src[c];
  }
  __syncthreads();
  fft_plane<-1>(x, az, Lz, npair, twz);
  // separate the line pairs, A_k = (Z_k + conj Z_{n-k}) / 2, B_k = (Z_k - conj Z_{n-k}) / 2i, and re-lay the plane as P[y][k]
  constexpr int KS = PLANE_ENTRIES / TPBP;
  double2 r[KS];
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    const int c = threadIdx.x + u * TPBP;
    if (c < n1 * n2h) {
      const int l = (int)(((float)c + 0.5f) * inv_nh), k = c - l * n2h;
      const int m = l >> 1;
      const double2 zk = x[k * Lz + m];
      const double2 zn = x[(k == 0 ? 0 : n2 - k) * Lz + m];
      if ((l & 1) == 0) r[u] = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
      else r[u] = make_double2(0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x));
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    const int c = threadIdx.x + u * TPBP;
    if (c < n1 * n2h) x[c] = r[u];
  }
  __syncthreads();
  fft_plane<-1>(x, ay, n2h, n2h, twy);
  for (int c = threadIdx.x; c < n1 * n2h; c += TPBP) dst[c] = x[c];
}

// inverse: in = half-complex planes, out = real planes (unnormalised: the 1/N sits in the kernel table)
__global__ __launch_bounds__(TPBP) void plane_inv_kernel(const double2* __restrict__ in, double* __restrict__ out, Axis az, Axis ay,
                                                         int Lz, int bufsz) {
  extern __shared__ double2 lds[];
  const int n2 = az.n, n1 = ay.n, n2h = n2 / 2 + 1, npair = (n1 + 1) / 2;
  double2* x = lds;
  double2* twz = lds + bufsz;
  double2* twy = twz + n2;
  for (int k = threadIdx.x; k < n2; k += TPBP) twz[k] = az.tw[k];
  for (int k = threadIdx.x; k < n1; k += TPBP) twy[k] = ay.tw[k];
  const double2* src = in + (int64_t)blockIdx.x * n1 * n2h;
  double* dst = out + (int64_t)blockIdx.x * n1 * n2;
  for (int c = threadIdx.x; c < n1 * n2h; c += TPBP) x[c] = src[c];
  __syncthreads();
  fft_plane<1>(x, ay, n2h, n2h, twy);
  // pairs of lines back into one complex line each: Z_k = A_k + i B_k, Z_{n-k} = conj(A_k) + i conj(B_k)
  constexpr int KS = PLANE_ENTRIES / TPBP / 2 + 1;
  const float inv_nh = 1.0f / (float)n2h, inv_n2 = 1.0f / (float)n2;
  double2 za[KS], zb[KS];
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    const int c = threadIdx.x + u * TPBP;
    if (c < npair * n2h) {
      const int m = (int)(((float)c + 0.5f) * inv_nh), k = c - m * n2h;
      double2 a = x[(2 * m) * n2h + k];
      double2 b = (2 * m + 1 < n1) ? x[(2 * m + 1) * n2h + k] : make_double2(0.0, 0.0);
      if (k == 0 || 2 * k == n2) { a.y = 0.0; b.y = 0.0; }
      za[u] = make_double2(a.x - b.y, a.y + b.x);
      zb[u] = make_double2(a.x + b.y, -a.y + b.x);
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    const int c = threadIdx.x + u * TPBP;
    if (c < npair * n2h) {
      const int m = (int)(((float)c + 0.5f) * inv_nh), k = c - m * n2h;
      x[k * Lz + m] = za[u];
      if (!(k == 0 || 2 * k == n2)) x[(n2 - k) * Lz + m] = zb[u];
    }
  }
  __syncthreads();
  fft_plane<1>(x, az, Lz, npair, twz);
  for (int c = threadIdx.x; c < n1 * n2; c += TPBP) {
    const int l = (int)(((float)c + 0.5f) * inv_n2), e = c - l * n2;
    dst[c] = ((const double*)(x + e * Lz + (l >> 1)))[l & 1];
  }
}

// ---- pipelined plane passes: ONE resident workgroup per CU walks over planes, the NEXT plane's global loads in flight (in
// registers) while the current plane goes through its stages in LDS.  A plane needs 115-117 KB of the CU's 160 KB, so a second
// workgroup cannot overlap its loads with the first one's arithmetic; a plane pass is then load + stages + store in sequence
// (26 us per 120^2 plane: about 12 us of memory time at the CU's share of the achievable bandwidth, the rest LDS round trips and
// butterflies).  768 threads (12 waves, 170 VGPRs each) hold the prefetched plane (NPRE values per thread) next to the stage
// registers without spilling - 1024 threads have 128 VGPRs and sit at 113-115 already.
// Everything about the plane is a compile-time constant (N x N points, two stages of radix RA and RB on both axes), so every
// per-thread array has exactly the size the plane needs.
template <int SIGN, int NT, int N, int RA, int RB, int MMAX = N / 2 + 1>
__device__ inline void fft_plane_fixed(double2* buf, int Ls, int M, const double2* tw, const int tid) {
  static_assert(RA * RB == N, "two stages");
  stage_plane<SIGN, RA, NT, ((N / RA) * MMAX + NT - 1) / NT>(buf, N, N, 1, Ls, M, tw, tid);
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  stage_plane<SIGN, RB, NT, ((N / RB) * MMAX + NT - 1) / NT>(buf, N, N / RA, RA, Ls, M, tw, tid2);
}

template <int NT, int N, int RA, int RB>
__global__ __launch_bounds__(NT) void plane_fwd_pipe_kernel(const double* __restrict__ in, double2* __restrict__ out, Axis az, Axis ay,
                                                            int Lz, int bufsz, int nplanes) {
  extern __shared__ double2 lds[];
  constexpr int n2 = N, n1 = N, n2h = N / 2 + 1, npair = (N + 1) / 2;
  constexpr int NPRE = (N * N + NT - 1) / NT;
  double2* x = lds;
  double2* twz = lds + bufsz;
  double2* twy = twz + n2;
  for (int k = threadIdx.x; k < n2; k += NT) twz[k] = az.tw[k];
  for (int k = threadIdx.x; k < n1; k += NT) twy[k] = ay.tw[k];
  constexpr int nreal = n1 * n2;
  const float inv_n2 = 1.0f / (float)n2, inv_nh = 1.0f / (float)n2h;
  double pre[NPRE];
  int pl = blockIdx.x;
  if (pl < nplanes) {
    const double* src = in + (int64_t)pl * nreal;
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int c = threadIdx.x + u * NT;
      pre[u] = c < nreal ? src[c] : 0.0;
    }
  }
  for (; pl < nplanes; pl += gridDim.x) {
    // an opaque copy of the thread index per round: index arithmetic that depends on it cannot be hoisted out of the plane loop
    // (hoisted, its per-element offsets were what the compiler spilled)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    if (n1 & 1)                                           // the last line has no partner: its imaginary slot is zero
      for (int e = tid; e < n2; e += NT) x[e * Lz + npair - 1].y = 0.0;
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int c = tid + u * NT;
      if (c < nreal) {
        const int l = (int)(((float)c + 0.5f) * inv_n2), e = c - l * n2;
        ((double*)(x + e * Lz + (l >> 1)))[l & 1] = pre[u];
      }
    }
    __syncthreads();
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // fresh opaque copy: this phase's index arithmetic stays in this phase
    const int nxt = pl + (int)gridDim.x;
    if (nxt < nplanes) {                                  // requested now, consumed at the top of the next round
      const double* src = in + (int64_t)nxt * nreal;
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        const int c = tid + u * NT;
        pre[u] = c < nreal ? src[c] : 0.0;
      }
    }
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // fresh opaque copy: this phase's index arithmetic stays in this phase
    fft_plane_fixed<-1, NT, N, RA, RB>(x, Lz, npair, twz, tid);
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // fresh opaque copy: this phase's index arithmetic stays in this phase
    constexpr int KS = (n1 * n2h + NT - 1) / NT;
    double2 r[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int c = tid + u * NT;
      if (c < n1 * n2h) {
        const int l = (int)(((float)c + 0.5f) * inv_nh), k = c - l * n2h;
        const int m = l >> 1;
        const double2 zk = x[k * Lz + m];
        const double2 zn = x[(k == 0 ? 0 : n2 - k) * Lz + m];
        if ((l & 1) == 0) r[u] = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
        else r[u] = make_double2(0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x));
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int c = tid + u * NT;
      if (c < n1 * n2h) x[c] = r[u];
    }
    __syncthreads();
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // fresh opaque copy: this phase's index arithmetic stays in this phase
    fft_plane_fixed<-1, NT, N, RA, RB>(x, n2h, n2h, twy, tid);
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // fresh opaque copy: this phase's index arithmetic stays in this phase
    double2* dst = out + (int64_t)pl * n1 * n2h;
    for (int c = tid; c < n1 * n2h; c += NT) dst[c] = x[c];
    __syncthreads();                                      // the next round overwrites the buffer
  }
}

template <int NT, int N, int RA, int RB>
__global__ __launch_bounds__(NT) void plane_inv_pipe_kernel(const double2* __restrict__ in, double* __restrict__ out, Axis az, Axis ay,
                                                            int Lz, int bufsz, int nplanes) {
  extern __shared__ double2 lds[];
  constexpr int n2 = N, n1 = N, n2h = N / 2 + 1, npair = (N + 1) / 2;
  constexpr int NPRE = (N * (N / 2 + 1) + NT - 1) / NT;
  double2* x = lds;
  double2* twz = lds + bufsz;
  double2* twy = twz + n2;
  for (int k = threadIdx.x; k < n2; k += NT) twz[k] = az.tw[k];
  for (int k = threadIdx.x; k < n1; k += NT) twy[k] = ay.tw[k];
  constexpr int ncplx = n1 * n2h;
  const float inv_nh = 1.0f / (float)n2h, inv_n2 = 1.0f / (float)n2;
  double2 pre[NPRE];
  int pl = blockIdx.x;
  if (pl < nplanes) {
    const double2* src = in + (int64_t)pl * ncplx;
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int c = threadIdx.x + u * NT;
      pre[u] = c < ncplx ? src[c] : make_double2(0.0, 0.0);
    }
  }
  for (; pl < nplanes; pl += gridDim.x) {
    // an opaque copy of the thread index per round: index arithmetic that depends on it cannot be hoisted out of the plane loop
    // (hoisted, its per-element offsets were what the compiler spilled)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int c = tid + u * NT;
      if (c < ncplx) x[c] = pre[u];
    }
    __syncthreads();
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // fresh opaque copy: this phase's index arithmetic stays in this phase
    const int nxt = pl + (int)gridDim.x;
    if (nxt < nplanes) {
      const double2* src = in + (int64_t)nxt * ncplx;
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        const int c = tid + u * NT;
        pre[u] = c < ncplx ? src[c] : make_double2(0.0, 0.0);
      }
    }
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // fresh opaque copy: this phase's index arithmetic stays in this phase
    fft_plane_fixed<1, NT, N, RA, RB>(x, n2h, n2h, twy, tid);
    // pairs of lines back into one complex line each: Z_k = A_k + i B_k, Z_{n-k} = conj(A_k) + i conj(B_k)
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // fresh opaque copy: this phase's index arithmetic stays in this phase
    constexpr int KS = (npair * n2h + NT - 1) / NT;
    double2 za[KS], zb[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int c = tid + u * NT;
      if (c < npair * n2h) {
        const int m = (int)(((float)c + 0.5f) * inv_nh), k = c - m * n2h;
        double2 a = x[(2 * m) * n2h + k];
        double2 b = (2 * m + 1 < n1) ? x[(2 * m + 1) * n2h + k] : make_double2(0.0, 0.0);
        if (k == 0 || 2 * k == n2) { a.y = 0.0; b.y = 0.0; }
        za[u] = make_double2(a.x - b.y, a.y + b.x);
        zb[u] = make_double2(a.x + b.y, -a.y + b.x);
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int c = tid + u * NT;
      if (c < npair * n2h) {
        const int m = (int)(((float)c + 0.5f) * inv_nh), k = c - m * n2h;
        x[k * Lz + m] = za[u];
        if (!(k == 0 || 2 * k == n2)) x[(n2 - k) * Lz + m] = zb[u];
      }
    }
    __syncthreads();
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // fresh opaque copy: this phase's index arithmetic stays in this phase
    fft_plane_fixed<1, NT, N, RA, RB>(x, Lz, npair, twz, tid);
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // fresh opaque copy: this phase's index arithmetic stays in this phase
    double* dst = out + (int64_t)pl * n1 * n2;
    for (int c = tid; c < n1 * n2; c += NT) {
      const int l = (int)(((float)c + 0.5f) * inv_n2), e = c - l * n2;
      dst[c] = ((const double*)(x + e * Lz + (l >> 1)))[l & 1];
    }
    __syncthreads();
  }
}

// ---- k-point form: real rows in, COMPLEX rows out (the kernel table of q != 0 is not inversion symmetric, so the product
// spectrum is not Hermitian).  Forward = the real-input half spectrum of the plane path + an x pass; then the half spectrum is
// expanded to the full one under the table multiply; inverse = x pass + one fused (y, z) complex plane pass writing Re and Im.
// full[b][kx][ky][kz] = tab[kx][ky][kz] * scale * F(kx, ky, kz),  F = half[...] for kz <= n2/2, conj(half[-kx, -ky, n2 - kz]) above
__global__ void expand_mul_kernel(const double2* __restrict__ half, double2* __restrict__ full, const double* __restrict__ tab,
                                  int n0, int n1, int n2, double scale) {
  const int n2h = n2 / 2 + 1;
  const int64_t G = (int64_t)n0 * n1 * n2, gc = (int64_t)n0 * n1 * n2h;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G) return;
  const int kz = (int)(idx % n2), ky = (int)((idx / n2) % n1), kx = (int)(idx / ((int64_t)n2 * n1));
  const double2* hb = half + (int64_t)blockIdx.y * gc;
  double2 v;
  if (kz < n2h) {
    v = hb[((int64_t)kx * n1 + ky) * n2h + kz];
  } else {
    const int mx = kx ? n0 - kx : 0, my = ky ? n1 - ky : 0;
    v = hb[((int64_t)mx * n1 + my) * n2h + (n2 - kz)];
    v.y = -v.y;
  }
  const double c = tab[idx] * scale;
  full[(int64_t)blockIdx.y * G + idx] = make_double2(c * v.x, c * v.y);
}

// inverse y and z transforms of one complex (y, z) plane in LDS, Re and Im written to separate real arrays
__global__ __launch_bounds__(TPBP) void plane_c2c_inv_kernel(const double2* __restrict__ in, double* __restrict__ out_re,
                                                             double* __restrict__ out_im, Axis az, Axis ay, int Lz, int bufsz) {
  extern __shared__ double2 lds[];
  const int n2 = az.n, n1 = ay.n;
  double2* x = lds;
  double2* twz = lds + bufsz;
  double2* twy = twz + n2;
  for (int k = threadIdx.x; k < n2; k += TPBP) twz[k] = az.tw[k];
  for (int k = threadIdx.x; k < n1; k += TPBP) twy[k] = ay.tw[k];
  const int64_t off = (int64_t)blockIdx.x * n1 * n2;
  const double2* src = in + off;
  for (int c = threadIdx.x; c < n1 * n2; c += TPBP) x[c] = src[c];            // P[y][z], pitch n2
  __syncthreads();
  fft_plane<1>(x, ay, n2, n2, twy);                                           // along y: element y, line z
  // transpose to X[z][y] (pitch Lz) through registers so that z becomes the element index
  constexpr int KS = PLANE_ENTRIES / TPBP;
  const float inv_n2 = 1.0f / (float)n2;
  double2 r[KS];
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    const int c = threadIdx.x + u * TPBP;
    const double2 v = x[c < n1 * n2 ? c : 0];
    r[u] = make_double2(v.x, v.y);                      // (component-wise: the struct copy kept the whole array in scratch)
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    const int c = threadIdx.x + u * TPBP;
    if (c < n1 * n2) {
      const int y = (int)(((float)c + 0.5f) * inv_n2), z = c - y * n2;
      x[z * Lz + y] = r[u];
    }
  }
  __syncthreads();
  fft_plane<1>(x, az, Lz, n1, twz);                                           // along z: element z, line y
  for (int c = threadIdx.x; c < n1 * n2; c += TPBP) {
    const int y = (int)(((float)c + 0.5f) * inv_n2), z = c - y * n2;
    const double2 v = x[z * Lz + y];
    out_re[off + c] = v.x;
    out_im[off + c] = v.y;
  }
}

// the same pass as a persistent workgroup with the next plane's loads in flight (see plane_fwd_pipe_kernel)
template <int NT, int N, int RA, int RB>
__global__ __launch_bounds__(NT) void plane_c2c_inv_pipe_kernel(const double2* __restrict__ in, double* __restrict__ out_re,
                                                                double* __restrict__ out_im, Axis az, Axis ay, int Lz, int bufsz,
                                                                int nplanes) {
  extern __shared__ double2 lds[];
  constexpr int n2 = N, n1 = N, nent = N * N;
  constexpr int NPRE = (nent + NT - 1) / NT;
  double2* x = lds;
  double2* twz = lds + bufsz;
  double2* twy = twz + n2;
  for (int k = threadIdx.x; k < n2; k += NT) twz[k] = az.tw[k];
  for (int k = threadIdx.x; k < n1; k += NT) twy[k] = ay.tw[k];
  const float inv_n2 = 1.0f / (float)n2;
  double2 pre[NPRE];
  int pl = blockIdx.x;
  if (pl < nplanes) {
    const double2* src = in + (int64_t)pl * nent;
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int c = threadIdx.x + u * NT;
      pre[u] = c < nent ? src[c] : make_double2(0.0, 0.0);
    }
  }
  for (; pl < nplanes; pl += gridDim.x) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int c = tid + u * NT;
      if (c < nent) x[c] = pre[u];                                               // P[y][z], pitch n2
    }
    __syncthreads();
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int nxt = pl + (int)gridDim.x;
    if (nxt < nplanes) {
      const double2* src = in + (int64_t)nxt * nent;
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        const int c = tid + u * NT;
        pre[u] = c < nent ? src[c] : make_double2(0.0, 0.0);
      }
    }
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    fft_plane_fixed<1, NT, N, RA, RB, N>(x, n2, n2, twy, tid);                    // along y: element y, line z
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    double2 r[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int c = tid + u * NT;
      const double2 v = x[c < nent ? c : 0];
      r[u] = make_double2(v.x, v.y);                    // (component-wise: the struct copy kept the whole array in scratch)
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int c = tid + u * NT;
      if (c < nent) {
        const int y = (int)(((float)c + 0.5f) * inv_n2), z = c - y * n2;
        x[z * Lz + y] = r[u];
      }
    }
    __syncthreads();
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    fft_plane_fixed<1, NT, N, RA, RB, N>(x, Lz, n1, twz, tid);                    // along z: element z, line y
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int64_t off = (int64_t)pl * nent;
    for (int c = tid; c < nent; c += NT) {
      const int y = (int)(((float)c + 0.5f) * inv_n2), z = c - y * n2;
      const double2 v = x[z * Lz + y];
      out_re[off + c] = v.x;
      out_im[off + c] = v.y;
    }
    __syncthreads();
  }
}

// geometry of the complex plane pass: 0 when the plane does not fit
size_t plane_c2c_lds_bytes(int n1, int n2, int* Lz_out, int* bufsz_out) {
  if (n1 * n2 > PLANE_ENTRIES) return 0;
  int Lz = n1;
  while (Lz % 16 != 1) ++Lz;
  const int bufsz = std::max(n2 * Lz, n1 * n2);
  const size_t bytes = sizeof(double2) * ((size_t)bufsz + n1 + n2);
  if (bytes > 160 * 1024) return 0;
  *Lz_out = Lz;
  *bufsz_out = bufsz;
  return bytes;
}

// plane path geometry: pitch of the z stage and LDS bytes, or 0 when the plane does not fit
size_t plane_lds_bytes(int n1, int n2, int* Lz_out, int* bufsz_out) {
  const int n2h = n2 / 2 + 1, npair = (n1 + 1) / 2;
  int Lz = npair;
  while (Lz % 16 != 1) ++Lz;
  const int bufsz = std::max(n2 * Lz, n1 * n2h);
  if (std::max(n2 * npair, n1 * n2h) > PLANE_ENTRIES || n1 * n2 > 2 * PLANE_ENTRIES) return 0;
  const size_t bytes = sizeof(double2) * ((size_t)bufsz + n1 + n2);
  if (bytes > 160 * 1024) return 0;
  *Lz_out = Lz;
  *bufsz_out = bufsz;
  return bytes;
}

template <class F>
void with_lines(int L, F f) {       // run f with the tile width as a compile-time constant
  switch (L) {
    case 16: f(std::integral_constant<int, 16>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    default: f(std::integral_constant<int, 1>{}); break;
  }
}

bool smooth235(const Axis& ax) {          // every stage radix is one the register butterflies know (2-3-5 smooth length)
  for (int i = 0; i < ax.nstage; ++i) {
    const int r = ax.radix[i];
    if (r == 7 || r == 11 || r == 13) return false;
  }
  return true;
}
int fast_lines(int n) {          // largest power of two <= 16 with n * lines <= 2048
  for (int L = 16; L >= 1; L >>= 1)
    if (n * L <= 2048) return L;
  return 0;
}

bool factorise(int n, Axis* ax) {
  ax->n = n;
  ax->nstage = 0;
  {
    // 2-3-5 smooth lengths (the FAST and PLANE paths): as few stages as the in-register radices up to 16 allow, the split with
    // the smallest largest radix among the shortest ones (120 = 12 x 10, 108 = 12 x 9, 128 = 16 x 8, 96 = 12 x 8)
    int m = n;
    for (int p : {2, 3, 5})
      while (m % p == 0) m /= p;
    static const int big[] = {16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2};
    static const bool big_radix = getenv("ISDF_FFT_BIG_RADIX") ? atoi(getenv("ISDF_FFT_BIG_RADIX")) != 0 : true;
    if (m == 1 && n > 1 && big_radix) {
      int best[MAXSTAGE], nbest = MAXSTAGE + 1, bestmax = 1 << 30;
      int cur[MAXSTAGE];
      // depth-first over non-increasing radix sequences (tiny search: n <= 1024)
      struct Rec {
        static void go(int rest, int maxr, int depth, int* cur, int* best, int& nbest, int& bestmax) {
          if (rest == 1) {
            const int mx = depth ? cur[0] : 1;
            if (depth < nbest || (depth == nbest && mx < bestmax)) {
              nbest = depth; bestmax = mx;
              for (int i = 0; i < depth; ++i) best[i] = cur[i];
            }
            return;
          }
          if (depth + 1 > nbest || depth >= MAXSTAGE) return;
          for (int r : big)
            if (r <= maxr && rest % r == 0) { cur[depth] = r; go(rest / r, r, depth + 1, cur, best, nbest, bestmax); }
        }
      };
      Rec::go(n, 16, 0, cur, best, nbest, bestmax);
      if (nbest <= MAXSTAGE) {
        ax->nstage = nbest;
        for (int i = 0; i < nbest; ++i) ax->radix[i] = best[i];
        return true;
      }
    }
  }
  const int order[4] = {4, 2, 3, 5};
  for (int r : order)
    while (n % r == 0 && n > 1) {
      if (ax->nstage >= MAXSTAGE) return false;
      ax->radix[ax->nstage++] = r;
      n /= r;
    }
  for (int p = 7; n > 1 && p <= 13; p += 2)
    while (n % p == 0) {
      if (ax->nstage >= MAXSTAGE) return false;
      ax->radix[ax->nstage++] = p;
      n /= p;
    }
  return n == 1;
}

int lines_per_tile(int n, int pad) {
  // two LDS buffers of n x (L + pad) complex numbers within 64 KB (two workgroups per CU)
  for (int L = 16; L >= 1; L >>= 1)
    if ((size_t)2 * n * (L + pad) * sizeof(double2) <= 64 * 1024) return L;
  return 0;
}

}  // namespace

// Whether conv_rows_own can take this mesh and this many rows at once (the batch-dependent launch limits it checks itself:
// callers fall back to hipFFT instead of failing).
bool conv_rows_own_supported(const int32_t mesh[3], int nb) {
  if (nb < 1 || (int64_t)nb * mesh[0] * mesh[1] * 4 >= 2147483647LL || (int64_t)nb * mesh[1] * (mesh[2] / 2 + 1) >= 2147483647LL)
    return false;
  for (int d = 0; d < 3; ++d) {
    Axis ax;
    if (mesh[d] < 1 || mesh[d] > 1024 || !factorise(mesh[d], &ax)) return false;
    if (lines_per_tile(mesh[d], d == 2 ? 1 : 0) < 1) return false;
  }
  return true;
}

// d_out (nb rows of G reals) = ifft(cg * fft(d_in rows)) with cg the scaled half-spectrum table of coulomb.hip
// (n0, n1, n2/2+1); zbuf: nb * gc complex scratch.  In place (d_out == d_in) allowed.
int conv_rows_own(isdf_handle h, const double* d_in, double* d_out, int nb, const int32_t mesh[3], const double* cg,
                  double2* zbuf) {
  const int n0 = mesh[0], n1 = mesh[1], n2 = mesh[2], n2h = n2 / 2 + 1;
  Axis ax[3];
  for (int d = 0; d < 3; ++d) {
    if (!factorise(mesh[d], &ax[d])) return isdf_fail(h, ISDF_ERR_ARG, "conv_rows_own: unsupported mesh dimension %d", mesh[d]);
    char name[32];
    snprintf(name, sizeof(name), "fft_tw_%d", mesh[d]);
    const bool fresh = h->ws.find(name) == h->ws.end();
    double2* tw = (double2*)isdf_ws(h, name, sizeof(double2) * (size_t)mesh[d]);
    if (!tw) return ISDF_ERR_HIP;
    if (fresh) {
      std::vector<double2> host(mesh[d]);
      for (int k = 0; k < mesh[d]; ++k) {
        const double t = -2.0 * 3.14159265358979323846 * (double)k / (double)mesh[d];
        host[k] = make_double2(cos(t), sin(t));
      }
      HIP_TRY(h, hipMemcpyAsync(tw, host.data(), sizeof(double2) * (size_t)mesh[d], hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    ax[d].tw = tw;
  }
  const int64_t gc = (int64_t)n0 * n1 * n2h;
  const int64_t nlines = (int64_t)nb * n0 * n1;
  ARG_CHECK(h, nlines * 4 < 2147483647LL && (int64_t)nb * cdiv((int64_t)n1 * n2h, 1) < 2147483647LL);
  const int LP = lines_per_tile(n2, 1);
  const int ZY = lines_per_tile(n1, 0), ZX = lines_per_tile(n0, 0);
  if (LP < 1 || ZY < 1 || ZX < 1) return isdf_fail(h, ISDF_ERR_ARG, "conv_rows_own: mesh too large for the LDS tiles");
  const int64_t G = (int64_t)n0 * n1 * n2;
  const int FZ = fast_lines(n2), FY = fast_lines(n1), FX = fast_lines(n0);
  int Lz = 0, bufsz = 0;
  const size_t plane_lds = (h->own_fft == 2 && smooth235(ax[0]) && smooth235(ax[1]) && smooth235(ax[2]) && FX >= 1)
                               ? plane_lds_bytes(n1, n2, &Lz, &bufsz) : 0;
  ProfScope ps(h, plane_lds ? "coulomb_conv_own_3pass[byte]" : "coulomb_conv_own_5pass[byte]", 32.0 * (double)G * nb,
               plane_lds ? 3 : 5);
  if (plane_lds) {
    // PLANE path: (z, y) forward per plane, x forward . table . x inverse, (y, z) inverse per plane
    // (set on every call: the attribute is per device, a process-wide flag would skip a second device)
    HIP_TRY(h, hipFuncSetAttribute((const void*)plane_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute((const void*)plane_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define ISDF_PIPE_ATTR(NN, RA, RB)                                                                                              \
  HIP_TRY(h, hipFuncSetAttribute((const void*)plane_fwd_pipe_kernel<PIPE_NT, NN, RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  HIP_TRY(h, hipFuncSetAttribute((const void*)plane_inv_pipe_kernel<PIPE_NT, NN, RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (n1 == n2) switch (n1) {
        case 64: ISDF_PIPE_ATTR(64, 8, 8) break;
        case 72: ISDF_PIPE_ATTR(72, 9, 8) break;
        case 80: ISDF_PIPE_ATTR(80, 10, 8) break;
        case 96: ISDF_PIPE_ATTR(96, 12, 8) break;
        case 100: ISDF_PIPE_ATTR(100, 10, 10) break;
        case 108: ISDF_PIPE_ATTR(108, 12, 9) break;
        case 120: ISDF_PIPE_ATTR(120, 12, 10) break;
        default: break;
      }
#undef ISDF_PIPE_ATTR
    const size_t ldsx = sizeof(double2) * ((size_t)n0 * FX + n0);
    const int ntx = (int)cdiv((int64_t)n1 * n2h, FX);
    hipStream_t st = h->stream;
    // Sub-batches sized so that the half spectra of a sub-batch stay in the 256 MB Infinity Cache between the three passes
    // (option "conv_sub_rows": rows per sub-batch; 0 = the whole batch at once, the round-2 behaviour): the middle pass then
    // reads and writes cache-resident lines and the inverse plane pass reads them back from there - HBM sees the real rows in
    // and out only.  Every sub-batch still launches >= 256 workgroups (rows x n0 planes).
    const int sub = h->conv_sub_rows > 0 ? std::min(nb, h->conv_sub_rows) : nb;
    for (int r0 = 0; r0 < nb; r0 += sub) {
      const int nr = std::min(sub, nb - r0);
      const dim3 gp((unsigned)((int64_t)nr * n0)), gx((unsigned)((int64_t)nr * ntx));
      double2* zb = zbuf + (int64_t)r0 * gc;
      const int nplanes = (int)((int64_t)nr * n0);
      const dim3 gpipe((unsigned)std::min<int64_t>(nplanes, h->num_cu));
      const double* src = d_in + (int64_t)r0 * G;
      double* dst = d_out + (int64_t)r0 * G;
      bool piped = false;
      // the pipelined passes exist for square planes of the sizes below (radices = what factorise() picks for them)
#define ISDF_PIPE_CASE(NN, RA, RB, FWD)                                                                                        \
  case NN:                                                                                                                      \
    if (ax[1].nstage == 2 && ax[1].radix[0] == RA && ax[1].radix[1] == RB) {                                                    \
      if (FWD) plane_fwd_pipe_kernel<PIPE_NT, NN, RA, RB><<<gpipe, dim3(PIPE_NT), plane_lds, st>>>(src, zb, ax[2], ax[1], Lz, bufsz, nplanes); \
      else plane_inv_pipe_kernel<PIPE_NT, NN, RA, RB><<<gpipe, dim3(PIPE_NT), plane_lds, st>>>(zb, dst, ax[2], ax[1], Lz, bufsz, nplanes);     \
      piped = true;                                                                                                             \
    }                                                                                                                           \
    break;
#define ISDF_PIPE_SWITCH(FWD)                                                                                                  \
  if (h->conv_pipe != 0 && n1 == n2) switch (n1) {                                                                              \
      ISDF_PIPE_CASE(64, 8, 8, FWD) ISDF_PIPE_CASE(72, 9, 8, FWD) ISDF_PIPE_CASE(80, 10, 8, FWD) ISDF_PIPE_CASE(96, 12, 8, FWD)  \
      ISDF_PIPE_CASE(100, 10, 10, FWD) ISDF_PIPE_CASE(108, 12, 9, FWD) ISDF_PIPE_CASE(120, 12, 10, FWD) default: break;          \
    }
      ISDF_PIPE_SWITCH(true)
      if (!piped) plane_fwd_kernel<<<gp, dim3(TPBP), plane_lds, st>>>(src, zb, ax[2], ax[1], Lz, bufsz);
      with_lines(FX, [&](auto z) {
        strided_fft_fast_kernel<2, decltype(z)::value><<<gx, dim3(TPB), ldsx, st>>>(zb, gc, (int64_t)n1 * n2h, n1 * n2h, ax[0], ntx, cg);
      });
      piped = false;
      ISDF_PIPE_SWITCH(false)
      if (!piped) plane_inv_kernel<<<gp, dim3(TPBP), plane_lds, st>>>(zb, dst, ax[2], ax[1], Lz, bufsz);
#undef ISDF_PIPE_SWITCH
#undef ISDF_PIPE_CASE
    }
    KERNEL_CHECK(h);
    return ISDF_OK;
  }
  if (smooth235(ax[0]) && smooth235(ax[1]) && smooth235(ax[2]) && FZ >= 1 && FY >= 1 && FX >= 1) {
    const size_t ldsz = sizeof(double2) * ((size_t)n2 * (FZ + 1) + n2);
    const size_t ldsy = sizeof(double2) * ((size_t)n1 * FY + n1), ldsx = sizeof(double2) * ((size_t)n0 * FX + n0);
    const int nty = (int)cdiv(n2h, FY), ntx = (int)cdiv((int64_t)n1 * n2h, FX);
    const dim3 gz((unsigned)cdiv(nlines, 2 * FZ)), gy((unsigned)((int64_t)nb * n0 * nty)), gx((unsigned)((int64_t)nb * ntx));
    hipStream_t st = h->stream;
    with_lines(FZ, [&](auto z) {
      z_r2c_fast_kernel<decltype(z)::value><<<gz, dim3(TPB), ldsz, st>>>(d_in, zbuf, nlines, ax[2]);
    });
    with_lines(FY, [&](auto z) {
      strided_fft_fast_kernel<0, decltype(z)::value><<<gy, dim3(TPB), ldsy, st>>>(zbuf, (int64_t)n1 * n2h, (int64_t)n2h, n2h, ax[1],
                                                                                 nty, (const double*)nullptr);
    });
    with_lines(FX, [&](auto z) {
      strided_fft_fast_kernel<2, decltype(z)::value><<<gx, dim3(TPB), ldsx, st>>>(zbuf, gc, (int64_t)n1 * n2h, n1 * n2h, ax[0], ntx, cg);
    });
    with_lines(FY, [&](auto z) {
      strided_fft_fast_kernel<1, decltype(z)::value><<<gy, dim3(TPB), ldsy, st>>>(zbuf, (int64_t)n1 * n2h, (int64_t)n2h, n2h, ax[1],
                                                                                 nty, (const double*)nullptr);
    });
    with_lines(FZ, [&](auto z) {
      z_c2r_fast_kernel<decltype(z)::value><<<gz, dim3(TPB), ldsz, st>>>(zbuf, d_out, nlines, ax[2]);
    });
    KERNEL_CHECK(h);
    return ISDF_OK;
  }
  // GENERIC path (a factor 7, 11 or 13): two LDS buffers
  {
    const size_t lds = (size_t)2 * n2 * (LP + 1) * sizeof(double2);
    hipLaunchKernelGGL(z_r2c_kernel, dim3((unsigned)cdiv(nlines, 2 * LP)), dim3(TPB), lds, h->stream, d_in, zbuf, nlines,
                       ax[2], LP);
  }
  {
    const size_t lds = (size_t)2 * n1 * ZY * sizeof(double2);
    const int nt = (int)cdiv(n2h, ZY);
    hipLaunchKernelGGL(strided_fft_kernel<0>, dim3((unsigned)((int64_t)nb * n0 * nt)), dim3(TPB), lds,
                       h->stream, zbuf, (int64_t)n1 * n2h, (int64_t)n2h, n2h, ax[1], ZY, nt, (const double*)nullptr);
  }
  {
    const size_t lds = (size_t)2 * n0 * ZX * sizeof(double2);
    const int nt = (int)cdiv((int64_t)n1 * n2h, ZX);
    hipLaunchKernelGGL(strided_fft_kernel<2>, dim3((unsigned)((int64_t)nb * nt)), dim3(TPB), lds,
                       h->stream, zbuf, gc, (int64_t)n1 * n2h, n1 * n2h, ax[0], ZX, nt, cg);
  }
  {
    const size_t lds = (size_t)2 * n1 * ZY * sizeof(double2);
    const int nt = (int)cdiv(n2h, ZY);
    hipLaunchKernelGGL(strided_fft_kernel<1>, dim3((unsigned)((int64_t)nb * n0 * nt)), dim3(TPB), lds,
                       h->stream, zbuf, (int64_t)n1 * n2h, (int64_t)n2h, n2h, ax[1], ZY, nt, (const double*)nullptr);
  }
  {
    const size_t lds = (size_t)2 * n2 * (LP + 1) * sizeof(double2);
    hipLaunchKernelGGL(z_c2r_kernel, dim3((unsigned)cdiv(nlines, 2 * LP)), dim3(TPB), lds, h->stream, zbuf, d_out, nlines,
                       ax[2], LP);
  }
  KERNEL_CHECK(h);
  return ISDF_OK;
}

// ---- spectral rows: X[r][2j], X[r][2j+1] = scale[j] * (Re, Im) fft(rows[r])[idx[j]] over a list of half-spectrum points.
// With scale^2 = multiplicity * w * coulG / G on the points inside a sphere, W = w conv(rows) rows^T = X X^T: the convolution's
// inverse transform and half of the P^2 G product's K dimension are never computed ("spectral W").
__global__ void spectral_pack_kernel(const double2* __restrict__ z, int64_t gc, const int32_t* __restrict__ idx,
                                     const double* __restrict__ scale, int npts, double2* __restrict__ out, int64_t ldx2) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ldx2) return;
  const int row = blockIdx.y;
  double2 v = make_double2(0.0, 0.0);                    // padding up to the leading dimension is zero
  if (j < npts) {
    const double2 t = z[(int64_t)row * gc + idx[j]];
    const double s = scale[j];
    v = make_double2(t.x * s, t.y * s);
  }
  out[(int64_t)row * ldx2 + j] = v;
}

bool spectral_rows_own_supported(isdf_handle h, const int32_t mesh[3], int nb) {
  if (h->own_fft != 2 || !conv_rows_own_supported(mesh, nb)) return false;
  Axis ax[3];
  for (int d = 0; d < 3; ++d)
    if (!factorise(mesh[d], &ax[d]) || !smooth235(ax[d])) return false;
  int a = 0, b = 0;
  return fast_lines(mesh[0]) >= 1 && plane_lds_bytes(mesh[1], mesh[2], &a, &b) != 0;
}

// d_out (nb rows, leading dimension ldx doubles, ldx even and >= 2 npts) from nb real rows of G points; zbuf: nb * gc complex
int spectral_rows_own(isdf_handle h, const double* d_in, int nb, const int32_t mesh[3], const int32_t* d_idx,
                      const double* d_scale, int npts, double* d_out, int64_t ldx, double2* zbuf) {
  const int n0 = mesh[0], n1 = mesh[1], n2 = mesh[2], n2h = n2 / 2 + 1;
  Axis ax[3];
  for (int d = 0; d < 3; ++d) {
    if (!factorise(mesh[d], &ax[d])) return isdf_fail(h, ISDF_ERR_ARG, "spectral_rows_own: unsupported mesh dimension %d", mesh[d]);
    char name[32];
    snprintf(name, sizeof(name), "fft_tw_%d", mesh[d]);
    const bool fresh = h->ws.find(name) == h->ws.end();
    double2* tw = (double2*)isdf_ws(h, name, sizeof(double2) * (size_t)mesh[d]);
    if (!tw) return ISDF_ERR_HIP;
    if (fresh) {
      std::vector<double2> host(mesh[d]);
      for (int k = 0; k < mesh[d]; ++k) {
        const double t = -2.0 * 3.14159265358979323846 * (double)k / (double)mesh[d];
        host[k] = make_double2(cos(t), sin(t));
      }
      HIP_TRY(h, hipMemcpyAsync(tw, host.data(), sizeof(double2) * (size_t)mesh[d], hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    ax[d].tw = tw;
  }
  const int64_t gc = (int64_t)n0 * n1 * n2h;
  int Lz = 0, bufsz = 0;
  const size_t lds_r = plane_lds_bytes(n1, n2, &Lz, &bufsz);
  const int FX = fast_lines(n0);
  ARG_CHECK(h, lds_r && FX >= 1 && (int64_t)nb * n0 < 2147483647LL && nb <= 65535 && (ldx & 1) == 0 && ldx >= 2 * (int64_t)npts);
  HIP_TRY(h, hipFuncSetAttribute((const void*)plane_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // algorithmic bytes: the real rows in, the packed rows out
  ProfScope ps(h, "spectral_rows_own[byte]", (8.0 * (double)n0 * n1 * n2 + 8.0 * (double)ldx) * nb, 3);
  hipStream_t st = h->stream;
  const size_t ldsx = sizeof(double2) * ((size_t)n0 * FX + n0);
  const dim3 gp((unsigned)((int64_t)nb * n0));
  const int nplanes = (int)((int64_t)nb * n0);
  const dim3 gpipe((unsigned)std::min<int64_t>(nplanes, h->num_cu));
  bool piped = false;
#define ISDF_PIPES_CASE(NN, RA, RB)                                                                                                \
  case NN:                                                                                                                          \
    if (ax[1].nstage == 2 && ax[1].radix[0] == RA && ax[1].radix[1] == RB) {                                                        \
      HIP_TRY(h, hipFuncSetAttribute((const void*)plane_fwd_pipe_kernel<PIPE_NT, NN, RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      plane_fwd_pipe_kernel<PIPE_NT, NN, RA, RB><<<gpipe, dim3(PIPE_NT), lds_r, st>>>(d_in, zbuf, ax[2], ax[1], Lz, bufsz, nplanes); \
      piped = true;                                                                                                                 \
    }                                                                                                                               \
    break;
  if (h->conv_pipe != 0 && n1 == n2) switch (n1) {
      ISDF_PIPES_CASE(64, 8, 8) ISDF_PIPES_CASE(72, 9, 8) ISDF_PIPES_CASE(80, 10, 8) ISDF_PIPES_CASE(96, 12, 8)
      ISDF_PIPES_CASE(100, 10, 10) ISDF_PIPES_CASE(108, 12, 9) ISDF_PIPES_CASE(120, 12, 10) default: break;
    }
#undef ISDF_PIPES_CASE
  if (!piped) plane_fwd_kernel<<<gp, dim3(TPBP), lds_r, st>>>(d_in, zbuf, ax[2], ax[1], Lz, bufsz);
  const int ntx = (int)cdiv((int64_t)n1 * n2h, FX);
  with_lines(FX, [&](auto z) {
    strided_fft_fast_kernel<0, decltype(z)::value><<<dim3((unsigned)((int64_t)nb * ntx)), dim3(TPB), ldsx, st>>>(
        zbuf, gc, (int64_t)n1 * n2h, n1 * n2h, ax[0], ntx, (const double*)nullptr);
  });
  const int64_t ldx2 = ldx / 2;
  spectral_pack_kernel<<<dim3((unsigned)cdiv(ldx2, 256), (unsigned)nb), dim3(256), 0, st>>>(zbuf, gc, d_idx, d_scale, npts,
                                                                                           (double2*)d_out, ldx2);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

// k-point convolution of real rows with a FULL real kernel table (n0, n1, n2; no 1/G inside): Vre + i Vim = ifft(tab fft(rows)).
// zhalf: nb * n0 n1 (n2/2+1) complex scratch, zfull: nb * G complex scratch.  Supported: 2-3-5 smooth meshes whose real and
// complex (y, z) planes fit LDS (up to ~100^2 per plane: the k-point configs of BASELINE.json); the caller falls back to hipFFT.
bool conv_rows_q_own_supported(isdf_handle h, const int32_t mesh[3], int nb) {
  if (!h->own_fft) return false;
  if (nb < 1 || nb > 65535 || (int64_t)nb * mesh[0] >= 2147483647LL) return false;
  Axis ax[3];
  for (int d = 0; d < 3; ++d)
    if (mesh[d] < 2 || mesh[d] > 1024 || !factorise(mesh[d], &ax[d]) || !smooth235(ax[d])) return false;
  int a = 0, b = 0;
  return fast_lines(mesh[0]) >= 1 && plane_lds_bytes(mesh[1], mesh[2], &a, &b) && plane_c2c_lds_bytes(mesh[1], mesh[2], &a, &b);
}

int conv_rows_q_own(isdf_handle h, const double* d_in, double* d_re, double* d_im, int nb, const int32_t mesh[3],
                    const double* tab, double2* zhalf, double2* zfull) {
  const int n0 = mesh[0], n1 = mesh[1], n2 = mesh[2], n2h = n2 / 2 + 1;
  Axis ax[3];
  for (int d = 0; d < 3; ++d) {
    if (!factorise(mesh[d], &ax[d])) return isdf_fail(h, ISDF_ERR_ARG, "conv_rows_q_own: unsupported mesh dimension %d", mesh[d]);
    char name[32];
    snprintf(name, sizeof(name), "fft_tw_%d", mesh[d]);
    const bool fresh = h->ws.find(name) == h->ws.end();
    double2* tw = (double2*)isdf_ws(h, name, sizeof(double2) * (size_t)mesh[d]);
    if (!tw) return ISDF_ERR_HIP;
    if (fresh) {
      std::vector<double2> host(mesh[d]);
      for (int k = 0; k < mesh[d]; ++k) {
        const double t = -2.0 * 3.14159265358979323846 * (double)k / (double)mesh[d];
        host[k] = make_double2(cos(t), sin(t));
      }
      HIP_TRY(h, hipMemcpyAsync(tw, host.data(), sizeof(double2) * (size_t)mesh[d], hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    ax[d].tw = tw;
  }
  const int64_t G = (int64_t)n0 * n1 * n2, gc = (int64_t)n0 * n1 * n2h;
  int Lz = 0, bufsz = 0, Lc = 0, bufc = 0;
  const size_t lds_r = plane_lds_bytes(n1, n2, &Lz, &bufsz), lds_c = plane_c2c_lds_bytes(n1, n2, &Lc, &bufc);
  const int FX = fast_lines(n0);
  ARG_CHECK(h, lds_r && lds_c && FX >= 1 && (int64_t)nb * n0 < 2147483647LL && nb <= 65535);
  HIP_TRY(h, hipFuncSetAttribute((const void*)plane_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute((const void*)plane_c2c_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // algorithmic bytes as the hipFFT form's label counts them: 8 G in, 16 G out per row and the table
  ProfScope ps(h, "coulomb_conv_q_own[byte]", 64.0 * (double)G * nb, 5);
  hipStream_t st = h->stream;
  const size_t ldsx = sizeof(double2) * ((size_t)n0 * FX + n0);
  const dim3 gp((unsigned)((int64_t)nb * n0));
  const int nplanes = (int)((int64_t)nb * n0);
  const dim3 gpipe((unsigned)std::min<int64_t>(nplanes, h->num_cu));
  int piped = 0;          // 1: forward plane pass piped, 2: inverse too
#define ISDF_PIPEQ_CASE(NN, RA, RB)                                                                                                \
  case NN:                                                                                                                          \
    if (ax[1].nstage == 2 && ax[1].radix[0] == RA && ax[1].radix[1] == RB) {                                                        \
      HIP_TRY(h, hipFuncSetAttribute((const void*)plane_fwd_pipe_kernel<PIPE_NT, NN, RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      HIP_TRY(h, hipFuncSetAttribute((const void*)plane_c2c_inv_pipe_kernel<PIPE_NT, NN, RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      if (!inverse) plane_fwd_pipe_kernel<PIPE_NT, NN, RA, RB><<<gpipe, dim3(PIPE_NT), lds_r, st>>>(d_in, zhalf, ax[2], ax[1], Lz, bufsz, nplanes); \
      else plane_c2c_inv_pipe_kernel<PIPE_NT, NN, RA, RB><<<gpipe, dim3(PIPE_NT), lds_c, st>>>(zfull, d_re, d_im, ax[2], ax[1], Lc, bufc, nplanes); \
      piped = 1;                                                                                                                    \
    }                                                                                                                               \
    break;
  auto plane_pass = [&](bool inverse) -> int {
    piped = 0;
    if (h->conv_pipe != 0 && n1 == n2) switch (n1) {
        ISDF_PIPEQ_CASE(64, 8, 8) ISDF_PIPEQ_CASE(72, 9, 8) ISDF_PIPEQ_CASE(80, 10, 8) ISDF_PIPEQ_CASE(96, 12, 8)
        default: break;                 // (100^2: the complex plane's prefetch does not fit the registers)
      }
    if (!piped) {
      if (!inverse) plane_fwd_kernel<<<gp, dim3(TPBP), lds_r, st>>>(d_in, zhalf, ax[2], ax[1], Lz, bufsz);
      else plane_c2c_inv_kernel<<<gp, dim3(TPBP), lds_c, st>>>(zfull, d_re, d_im, ax[2], ax[1], Lc, bufc);
    }
    return ISDF_OK;
  };
#undef ISDF_PIPEQ_CASE
  { int rc = plane_pass(false); if (rc != ISDF_OK) return rc; }
  const int ntx = (int)cdiv((int64_t)n1 * n2h, FX), ntf = (int)cdiv((int64_t)n1 * n2, FX);
  with_lines(FX, [&](auto z) {
    strided_fft_fast_kernel<0, decltype(z)::value><<<dim3((unsigned)((int64_t)nb * ntx)), dim3(TPB), ldsx, st>>>(
        zhalf, gc, (int64_t)n1 * n2h, n1 * n2h, ax[0], ntx, (const double*)nullptr);
  });
  expand_mul_kernel<<<dim3((unsigned)cdiv(G, 256), (unsigned)nb), dim3(256), 0, st>>>(zhalf, zfull, tab, n0, n1, n2, 1.0 / (double)G);
  with_lines(FX, [&](auto z) {
    strided_fft_fast_kernel<1, decltype(z)::value><<<dim3((unsigned)((int64_t)nb * ntf)), dim3(TPB), ldsx, st>>>(
        zfull, G, (int64_t)n1 * n2, n1 * n2, ax[0], ntf, (const double*)nullptr);
  });
  { int rc = plane_pass(true); if (rc != ISDF_OK) return rc; }
  KERNEL_CHECK(h);
  return ISDF_OK;
}

// S2 (refined stage): pivoted Cholesky of an EXPLICIT symmetric positive semidefinite matrix A (m x m, HBM) —
// the pair-density Gram matrix restricted to a candidate set, A = (phi_c^T phi_c)^2.  Same pivot rule as
// select_ip.hip (pyscf/lib/scipy_helper.py:71-110 + the tie rule of mi355_isdf.h); the arithmetic is that of the
// reference's pivoted_cholesky_python applied to A itself, organised like LAPACK's dpstrf:
//
//   * right-looking in panels of nb pivots: after a panel, the trailing matrix is updated in place,
//     A <- A - Lp^T Lp, so a pivot column is one row of A minus the contributions of the CURRENT panel only —
//     8*m*(1 + j_in_panel) bytes per pivot instead of 8*m*j for the left-looking form (33 TB -> 0.6 TB at m = 33280,
//     P = 16640).  Only the BLOCK-LOWER part of A is kept up to date (column strips of SW columns, each updated from its
//     own first row down: half the flops of the full m x m x nb product); entry (p, i) of the symmetric matrix is read as
//     A[p][i] where row p lies in or below i's strip and as A[i][p] otherwise;
//   * ONE launch per pivot.  Every workgroup owns 256 columns.  It first finds the pivot redundantly from the
//     per-workgroup maxima of the previous step (nwg doubles) and the 256 residual diagonals of the first workgroup
//     within the tie tolerance, then stages the panel entries of the pivot (pl[t] = Lp[t, p]) in LDS and updates its own
//     columns: row = (A[p, i] - sum_t Lp[t, i] pl[t]) / sqrt(d_p), d[i] -= row^2.  The residual diagonal and the
//     workgroup maxima are double-buffered (read step j-1, write step j) so that no workgroup reads what another one
//     is writing in the same launch.
#include "common.h"
#include <cfloat>
#include <climits>

namespace {

constexpr int PANEL_MAX = 256; // most pivots per panel (rows of Lp whose pivot entries are staged in LDS)
constexpr int SW = 2048;       // strip width of the block-lower trailing update (a multiple of every workgroup width)

struct GramState {
  double tol;
  int rank;
  int done;
};

__device__ inline double wmax(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ inline int wmin(int v) {
  for (int o = 32; o > 0; o >>= 1) {
    int u = __shfl_xor(v, o);
    v = u < v ? u : v;
  }
  return v;
}

// d0[i] = A[i,i]; per-workgroup maxima
template <int TPB>
__global__ __launch_bounds__(TPB) void gram_diag_kernel(const double* __restrict__ A, int64_t ldA, int m,
                                                       double* __restrict__ d, double* __restrict__ wgmax) {
  __shared__ double red[TPB / 64];
  const int i = blockIdx.x * TPB + threadIdx.x;
  double v = 0.0;
  if (i < m) {
    v = A[(int64_t)i * ldA + i];
    d[i] = v;
  }
  double mx = wmax(i < m ? fmax(v, 0.0) : 0.0);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < TPB / 64; ++w) mx = fmax(mx, red[w]);
    wgmax[blockIdx.x] = mx;
  }
}

// One pivot: j = global pivot index, jl = index inside the current panel (rows of Lp).  TPB = columns per workgroup; the
// arithmetic of a column does not depend on it.  Measured at m = 39 936 (profiles/r03_gram_pivot_step_widths.log): a step costs
// 7 us + 0.04 us per panel row already made - the second term is the panel streaming out of the Infinity Cache at 7.4 TB/s, the
// first is the launch boundary, the dispatch of the workgroups and three dependent memory round trips.  Narrower workgroups are
// SLOWER (more of them to dispatch, each repeating the pivot search), so 256 stays the default.
template <int TPB>
__global__ __launch_bounds__(TPB) void gram_pivot_step_kernel(
    const double* __restrict__ A, int64_t ldA, int m, double* __restrict__ Lp, int64_t ldL,
    const double* __restrict__ d_old, double* __restrict__ d_new, const double* __restrict__ wg_old,
    double* __restrict__ wg_new, int nwg, int j, int jl, int nip, double tol_in, double tie_rtol,
    GramState* __restrict__ st, int64_t* __restrict__ piv) {
  constexpr int NW = TPB / 64;
  constexpr int CH = 32;           // loads of one chunk of the panel's rows, all issued before the chunk's first fma
  __shared__ double s_red[NW];
  __shared__ int s_redi[2][NW];
  __shared__ double s_pl[PANEL_MAX];
  __shared__ double s_dp;
  const int tid = threadIdx.x;
  const int wg = blockIdx.x;
  const int i = wg * TPB + tid;
  const bool valid = i < m;
  // The step is a chain of dependent memory round trips (maxima -> the first workgroup's diagonals -> the pivot's row and
  // panel entries -> the panel rows of the own column) on a chip where every wave sits alone on its SIMD (m / 64 waves, 1024
  // SIMDs): everything that does not depend on the pivot is requested up front - the own diagonal, the state, and the first
  // chunk of the own column's panel rows - and what depends only on the pivot (its row of A, its panel entries) goes out together.
  const double* __restrict__ pL = Lp + i;
  const double dold = valid ? d_old[i] : -1.0;
  const int st_done = st->done;
  const double st_tol = st->tol;
  double v0[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) v0[u] = (valid && u < jl) ? pL[(int64_t)u * ldL] : 0.0;

  // (1) global maximum of the residual diagonal from the per-workgroup maxima of the previous step, and (2) the first
  // workgroup whose maximum is within the tie tolerance (second sweep over the same, now cached, values)
  double gm = 0.0;
  for (int w = tid; w < nwg; w += TPB) gm = fmax(gm, wg_old[w]);
  gm = wmax(gm);
  if ((tid & 63) == 0) s_red[tid >> 6] = gm;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) gm = fmax(gm, s_red[w]);
  const double tol = (j == 0) ? (tol_in < 0 ? (double)m * DBL_EPSILON * gm : tol_in) : st_tol;
  const bool stop = st_done || j >= nip || !(gm > tol);
  if (stop) {
    if (valid) {
      Lp[(int64_t)jl * ldL + i] = 0.0;
      d_new[i] = dold;
    }
    if (tid == 0) {
      wg_new[wg] = wg_old[wg];
      if (wg == 0) { st->done = 1; if (j == 0) { st->tol = tol; st->rank = 0; } }
    }
    return;
  }
  const double thr = gm * (1.0 - tie_rtol);
  int wfirst = INT_MAX;
  for (int w = tid; w < nwg; w += TPB)
    if (wg_old[w] >= thr && wg_old[w] > 0.0) { wfirst = w; break; }
  wfirst = wmin(wfirst);
  if ((tid & 63) == 0) s_redi[0][tid >> 6] = wfirst;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) wfirst = s_redi[0][w] < wfirst ? s_redi[0][w] : wfirst;
  // the first column of that workgroup within the tolerance; the lane that holds it publishes sqrt(d_p)
  int cand = INT_MAX;
  double dv = 0.0;
  const int c = wfirst == INT_MAX ? m : wfirst * TPB + tid;
  if (c < m) {
    dv = d_old[c];
    if (dv >= thr && dv > 0.0) cand = c;
  }
  const int mine = cand;
  cand = wmin(cand);
  if ((tid & 63) == 0) s_redi[1][tid >> 6] = cand;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) cand = s_redi[1][w] < cand ? s_redi[1][w] : cand;
  const int p = cand;
  if (p == INT_MAX) {   // cannot happen while the maxima are consistent with d; never index with it
    if (valid) {
      Lp[(int64_t)jl * ldL + i] = 0.0;
      d_new[i] = dold;
    }
    if (tid == 0) {
      wg_new[wg] = wg_old[wg];
      if (wg == 0) st->done = 1;
    }
    return;
  }
  // (3) what depends on the pivot alone: its row of A at the own column (symmetric read from the block-lower part: (p, i) is
  // current iff p >= first row of i's strip) and its entries of the current panel
  const bool work = valid && i != p && !(dold < 0.0);
  double col = 0.0;
  if (work) col = (p >= (i / SW) * SW) ? A[(int64_t)p * ldA + i] : A[(int64_t)i * ldA + p];
  for (int t = tid; t < jl; t += TPB) s_pl[t] = Lp[(int64_t)t * ldL + p];
  if (mine == p) s_dp = sqrt(dv);
  __syncthreads();
  const double dp = s_dp;
  // (4) this workgroup's columns: one fixed fma chain per column (t ascending)
  double dnew = 0.0;
  if (valid) {
    double row;
    if (i == p) {
      row = dp;
      dnew = -1.0;
    } else if (dold < 0.0) {       // an earlier pivot: its residual row is exactly zero
      row = 0.0;
      dnew = -1.0;
    } else {
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (u < jl) col = fma(-v0[u], s_pl[u], col);
      int t = CH;
      for (; t + CH <= jl; t += CH) {
        double v[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) v[u] = pL[(int64_t)(t + u) * ldL];
#pragma unroll
        for (int u = 0; u < CH; ++u) col = fma(-v[u], s_pl[t + u], col);
      }
      if (t < jl) {
        double v[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) v[u] = (t + u < jl) ? pL[(int64_t)(t + u) * ldL] : 0.0;
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if (t + u < jl) col = fma(-v[u], s_pl[t + u], col);
      }
      row = col / dp;
      dnew = fma(-row, row, dold);
      if (dnew < 0.0) dnew = 0.0;
    }
    Lp[(int64_t)jl * ldL + i] = row;
    d_new[i] = dnew;
  }
  double mx = wmax(valid ? fmax(dnew, 0.0) : 0.0);
  if ((tid & 63) == 0) s_red[tid >> 6] = mx;     // s_red's readers of step (1) are all past the later barriers
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmax(mx, s_red[w]);
    wg_new[wg] = mx;
    if (wg == 0) {
      piv[j] = p;
      st->rank = j + 1;
      if (j == 0) st->tol = tol;
    }
  }
}

}  // namespace

extern "C" int isdf_select_ip_gram(isdf_handle h, double* d_A, int m, int64_t ldA, int nip, double tol,
                                   double tie_rtol, int panel, int64_t* d_piv, int32_t* rank) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_A && d_piv && rank);
  ARG_CHECK(h, m > 0 && ldA >= m && nip > 0 && tie_rtol >= 0.0 && tie_rtol < 1.0);
  if (panel <= 0) panel = 256;
  ARG_CHECK(h, panel <= PANEL_MAX);
  if (nip > m) nip = m;
  const int TPB = h->gram_pivot_tpb;
  const int nwg = (int)cdiv(m, TPB);
  const int64_t ldL = ((int64_t)m + 31) / 32 * 32;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t b_d = al(sizeof(double) * m), b_w = al(sizeof(double) * nwg), b_st = al(sizeof(GramState));
  const size_t b_L = al(sizeof(double) * (size_t)panel * ldL);
  char* ws = (char*)isdf_ws(h, "select_gram", 2 * b_d + 2 * b_w + b_st + b_L);
  if (!ws) return ISDF_ERR_HIP;
  double* d_d[2];
  double* d_w[2];
  d_d[0] = (double*)ws; ws += b_d;
  d_d[1] = (double*)ws; ws += b_d;
  d_w[0] = (double*)ws; ws += b_w;
  d_w[1] = (double*)ws; ws += b_w;
  GramState* d_st = (GramState*)ws; ws += b_st;
  double* d_Lp = (double*)ws;
  HIP_TRY(h, hipMemsetAsync(d_st, 0, sizeof(GramState), h->stream));
  HIP_TRY(h, hipMemsetAsync(d_piv, 0xff, sizeof(int64_t) * (size_t)nip, h->stream));
  if (TPB == 64) hipLaunchKernelGGL(gram_diag_kernel<64>, dim3(nwg), dim3(64), 0, h->stream, d_A, ldA, m, d_d[0], d_w[0]);
  else if (TPB == 128) hipLaunchKernelGGL(gram_diag_kernel<128>, dim3(nwg), dim3(128), 0, h->stream, d_A, ldA, m, d_d[0], d_w[0]);
  else hipLaunchKernelGGL(gram_diag_kernel<256>, dim3(nwg), dim3(256), 0, h->stream, d_A, ldA, m, d_d[0], d_w[0]);
  KERNEL_CHECK(h);
  int cur = 0;
  for (int k0 = 0; k0 < nip; k0 += panel) {
    const int nb = nip - k0 < panel ? nip - k0 : panel;
    {
      ProfScope ps(h, "gram_pivot_step_kernel[byte]", 8.0 * (double)m * (0.5 * nb * (nb - 1) + 4.0 * nb), nb);
      for (int jl = 0; jl < nb; ++jl) {
#define ISDF_GRAM_STEP(T)                                                                                              \
  hipLaunchKernelGGL(gram_pivot_step_kernel<T>, dim3(nwg), dim3(T), 0, h->stream, d_A, ldA, m, d_Lp, ldL, d_d[cur],     \
                     d_d[cur ^ 1], d_w[cur], d_w[cur ^ 1], nwg, k0 + jl, jl, nip, tol, tie_rtol, d_st, d_piv)
        if (TPB == 64) ISDF_GRAM_STEP(64);
        else if (TPB == 128) ISDF_GRAM_STEP(128);
        else ISDF_GRAM_STEP(256);
#undef ISDF_GRAM_STEP
        cur ^= 1;
      }
      KERNEL_CHECK(h);
    }
    // one host synchronisation per panel: at most `panel` launches are ever outstanding.  Unbounded, the ~20 000
    // back-to-back launches of a configs[2] selection overran rocprofv3's counter-collection path (SIGSEGV in the tool at the
    // first page past one of its buffers, profiles/r02_rocprofv3_pmc_abort_in_select_ip_gram.log); costs < 2 ms per build
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (k0 + nb < nip) {
      // trailing update A <- A - Lp^T Lp on the block-lower part: strip [c0, c1) from its first row down
      for (int c0 = 0; c0 < m; c0 += SW) {
        const int c1 = c0 + SW < m ? c0 + SW : m;
        int rc = gemm_rm(h, 'T', 'N', m - c0, c1 - c0, nb, -1.0, d_Lp + c0, ldL, d_Lp + c0, ldL, 1.0,
                         d_A + (int64_t)c0 * ldA + c0, ldA);
        if (rc != ISDF_OK) return rc;
      }
    }
  }
  GramState hst;
  HIP_TRY(h, hipMemcpyAsync(&hst, d_st, sizeof(GramState), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  *rank = hst.rank;
  return ISDF_OK;
}

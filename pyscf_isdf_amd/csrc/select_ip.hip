// S2: interpolation-point selection — pivoted Cholesky of the implicit pair-density Gram matrix
// A(r,r') = (sum_mu ao[mu,r] ao[mu,r'])^2, batched over independent column blocks of ao.
//
// Left-looking, one pivot per step for every block at once.  Per step three launches:
//   find_first   every workgroup scans its slice of the residual diagonal d for the LOWEST index with
//                d >= (1 - tie_rtol) * blockmax  -> atomicMin into the block's candidate slot
//   take_pivot   one workgroup per block: records the pivot, decides termination, gathers the pivot's
//                AO column pv[nao] and its previous Cholesky entries pl[j] into contiguous scratch,
//                re-arms the block's max/argmin slots
//   update       every lane owns one grid point i:  s = sum_mu ao[mu,i] pv[mu]  (coalesced along the
//                grid), col = s^2 - sum_{t<j} L[t,i] pl[t], L[j,i] = col / sqrt(d_p), d[i] -= L[j,i]^2,
//                and the workgroup's max of the new d goes to the block's slot by atomicMax
// pv/pl are staged in LDS (panel of the factorisation) and broadcast to the lanes.
// The stream of ao (nao x m) and of the growing L (j x m) is the HBM cost: 8*(nao + j)*m bytes/step.
// Pivot rule: pyscf/lib/scipy_helper.py:71-110 plus the deterministic tie rule of mi355_isdf.h.
#include "common.h"
#include <cfloat>
#include <climits>

namespace {

constexpr int TPB = 256;

struct BlkState {
  unsigned long long dmax_bits;   // max residual diagonal of the block (bit pattern of a double >= 0)
  long long cand;                 // lowest local index within tolerance of the max (LLONG_MAX = none)
  double dp;                      // sqrt(d[pivot]) of the current step
  double tol;
  long long pivot;                // local index of the current pivot
  int done;
  int rank;
};

__device__ inline double wave_max_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ inline long long wave_min_ll(long long v) {
  for (int o = 32; o > 0; o >>= 1) {
    long long u = __shfl_xor(v, o);
    v = u < v ? u : v;
  }
  return v;
}

// d[i] = (sum_mu ao[mu,i]^2)^2 ; block max -> state
__global__ __launch_bounds__(TPB) void init_diag_kernel(
    const double* __restrict__ ao, int nao, int64_t ld, const int* __restrict__ wg_blk,
    const int64_t* __restrict__ wg_lo, const int64_t* __restrict__ blk_off, double* __restrict__ d,
    BlkState* __restrict__ st) {
  __shared__ double red[TPB / 64];
  const int b = wg_blk[blockIdx.x];
  const int64_t i = wg_lo[blockIdx.x] + threadIdx.x;
  const bool valid = i < blk_off[b + 1];
  double s = 0.0;
  if (valid) {
    const double* __restrict__ p = ao + i;
#pragma unroll 8
    for (int mu = 0; mu < nao; ++mu) {
      const double v = p[(int64_t)mu * ld];
      s = fma(v, v, s);
    }
    s = s * s;
    d[i] = s;
  }
  double m = wave_max_d(valid ? s : 0.0);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < TPB / 64; ++w) m = fmax(m, red[w]);
    atomicMax(&st[b].dmax_bits, (unsigned long long)__double_as_longlong(m));
  }
}

__global__ __launch_bounds__(TPB) void find_first_kernel(
    const double* __restrict__ d, const int* __restrict__ wg_blk, const int64_t* __restrict__ wg_lo,
    const int64_t* __restrict__ blk_off, BlkState* __restrict__ st, double tie_rtol) {
  __shared__ long long red[TPB / 64];
  const int b = wg_blk[blockIdx.x];
  if (st[b].done) return;
  const int64_t i = wg_lo[blockIdx.x] + threadIdx.x;
  const double dmax = __longlong_as_double((long long)st[b].dmax_bits);
  const double thr = dmax * (1.0 - tie_rtol);
  long long c = LLONG_MAX;
  if (i < blk_off[b + 1] && d[i] >= thr && d[i] > 0.0) c = i - blk_off[b];
  c = wave_min_ll(c);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < TPB / 64; ++w) c = red[w] < c ? red[w] : c;
    if (c != LLONG_MAX) atomicMin(&st[b].cand, c);
  }
}

// one workgroup per block
__global__ __launch_bounds__(TPB) void take_pivot_kernel(
    const double* __restrict__ ao, int nao, int64_t ld, const double* __restrict__ L, int64_t ldL,
    const double* __restrict__ d, const int64_t* __restrict__ blk_off, const int* __restrict__ nip,
    int j, double tol_in, int kmax, BlkState* __restrict__ st, int64_t* __restrict__ piv,
    double* __restrict__ pv, double* __restrict__ pl, int nh, double* __restrict__ pvr) {
  const int b = blockIdx.x;
  BlkState s = st[b];
  if (s.done) return;
  const double dmax = __longlong_as_double((long long)s.dmax_bits);
  const int64_t m = blk_off[b + 1] - blk_off[b];
  double tol = s.tol;
  if (j == 0) tol = tol_in < 0 ? (double)m * DBL_EPSILON * dmax : tol_in;
  const bool stop = (j >= nip[b]) || !(dmax > tol) || s.cand == LLONG_MAX;
  __syncthreads();
  if (stop) {
    if (threadIdx.x == 0) { st[b].done = 1; st[b].tol = tol; }
    return;
  }
  const int64_t p = blk_off[b] + s.cand;
  for (int mu = threadIdx.x; mu < nao; mu += TPB) pv[(int64_t)b * nao + mu] = ao[(int64_t)mu * ld + p];
  if (nh > 0)   // rotated pivot vector [Im u_p ; -Re u_p] for the imaginary part of S
    for (int mu = threadIdx.x; mu < nao; mu += TPB)
      pvr[(int64_t)b * nao + mu] = (mu < nh) ? ao[(int64_t)(nh + mu) * ld + p] : -ao[(int64_t)(mu - nh) * ld + p];
  for (int t = threadIdx.x; t < j; t += TPB) pl[(int64_t)b * kmax + t] = L[(int64_t)t * ldL + p];
  if (threadIdx.x == 0) {
    piv[(int64_t)b * kmax + j] = s.cand;
    st[b].pivot = s.cand;
    st[b].dp = sqrt(d[p]);
    st[b].tol = tol;
    st[b].rank = j + 1;
    st[b].dmax_bits = 0ull;
    st[b].cand = LLONG_MAX;
  }
}

template <bool LDS_PANEL>
__global__ __launch_bounds__(TPB) void update_kernel(
    const double* __restrict__ ao, int nao, int64_t ld, double* __restrict__ L, int64_t ldL,
    double* __restrict__ d, const int* __restrict__ wg_blk, const int64_t* __restrict__ wg_lo,
    const int64_t* __restrict__ blk_off, int j, int kmax, BlkState* __restrict__ st,
    const double* __restrict__ pv, const double* __restrict__ pl, int nh, const double* __restrict__ pvr) {
  // nh > 0: complex mode, rows [0,nh) = Re u, rows [nh,2nh) = Im u (nao = 2 nh):
  //   S(p,i) = sum conj(u_p) u_i,  Re S = sum_m X[m,i] pv[m],  Im S = sum_m X[m,i] pvr[m],
  //   pvr = [Im u_p ; -Re u_p];  Gram entry = Re^2 + Im^2.
  extern __shared__ double panel[];   // pv[nao] | pl[j] (| pvr[nao] in complex mode)
  __shared__ double red[TPB / 64];
  const int b = wg_blk[blockIdx.x];
  if (st[b].done) return;
  // the pivot panel (pv | pl | pvr) is broadcast to all lanes: from LDS when it fits, otherwise straight
  // from global memory (wave-uniform addresses, L2 resident) — the k-point mode with many AOs
  const double* s_pv;
  const double* s_pl;
  const double* s_pvr;
  if (LDS_PANEL) {
    double* w_pv = panel;
    double* w_pl = panel + nao;
    double* w_pvr = panel + nao + j;
    for (int mu = threadIdx.x; mu < nao; mu += TPB) w_pv[mu] = pv[(int64_t)b * nao + mu];
    for (int t = threadIdx.x; t < j; t += TPB) w_pl[t] = pl[(int64_t)b * kmax + t];
    if (nh > 0)
      for (int mu = threadIdx.x; mu < nao; mu += TPB) w_pvr[mu] = pvr[(int64_t)b * nao + mu];
    __syncthreads();
    s_pv = w_pv; s_pl = w_pl; s_pvr = w_pvr;
  } else {
    s_pv = pv + (int64_t)b * nao;
    s_pl = pl + (int64_t)b * kmax;
    s_pvr = pvr + (int64_t)b * nao;
  }
  const int64_t i = wg_lo[blockIdx.x] + threadIdx.x;
  const bool valid = i < blk_off[b + 1];
  double dnew = 0.0;
  if (valid) {
    const double dp = st[b].dp;
    const int64_t ploc = blk_off[b] + st[b].pivot;
    const double dold = d[i];
    double row;
    if (i == ploc) {
      row = dp;
      dnew = -1.0;
    } else if (dold < 0.0) {       // an earlier pivot: residual row is exactly zero
      row = 0.0;
      dnew = -1.0;
    } else {
      const double* __restrict__ pa = ao + i;
      double s0 = 0.0, s1 = 0.0;
      if (nh > 0) {
#pragma unroll 8
        for (int mu = 0; mu < nao; ++mu) {
          const double x = pa[(int64_t)mu * ld];
          s0 = fma(x, s_pv[mu], s0);
          s1 = fma(x, s_pvr[mu], s1);
        }
      } else {
#pragma unroll 8
        for (int mu = 0; mu < nao; ++mu) s0 = fma(pa[(int64_t)mu * ld], s_pv[mu], s0);
      }
      double col = fma(s1, s1, s0 * s0);
      const double* __restrict__ pL = L + i;
#pragma unroll 8
      for (int t = 0; t < j; ++t) col = fma(-pL[(int64_t)t * ldL], s_pl[t], col);
      row = col / dp;
      dnew = fma(-row, row, dold);
      if (dnew < 0.0) dnew = 0.0;   // rounding can push an exhausted point below zero; keep it alive-but-empty
    }
    L[(int64_t)j * ldL + i] = row;
    d[i] = dnew;
  }
  double m = wave_max_d(valid ? fmax(dnew, 0.0) : 0.0);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < TPB / 64; ++w) m = fmax(m, red[w]);
    atomicMax(&st[b].dmax_bits, (unsigned long long)__double_as_longlong(m));
  }
}

}  // namespace

namespace {
// owner[g] = the atom nearest to grid point g under the minimum-image convention (the 27 neighbouring images of every atom);
// among atoms within tie_atol of the smallest distance the LOWEST index wins (symmetric crystals put many points exactly
// between atoms).  Atom positions and the 27 lattice shifts sit in LDS; one grid point per lane, two sweeps over the atoms.
__global__ __launch_bounds__(256) void partition_kernel(const double* __restrict__ coords, int64_t G, const double* __restrict__ atoms,
                                                        int natm, const double* __restrict__ shifts, double tie_atol,
                                                        int32_t* __restrict__ owner) {
  extern __shared__ double sm[];            // atoms (3 natm) | shifts (81)
  double* s_at = sm;
  double* s_sh = sm + 3 * natm;
  for (int t = threadIdx.x; t < 3 * natm; t += 256) s_at[t] = atoms[t];
  for (int t = threadIdx.x; t < 81; t += 256) s_sh[t] = shifts[t];
  __syncthreads();
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  const double x = coords[g], y = coords[G + g], z = coords[2 * G + g];
  double dmin = 1e300;
  for (int a = 0; a < natm; ++a) {
    const double ax = x - s_at[3 * a], ay = y - s_at[3 * a + 1], az = z - s_at[3 * a + 2];
    double d2 = 1e300;
    for (int t = 0; t < 27; ++t) {
      const double dx = ax - s_sh[3 * t], dy = ay - s_sh[3 * t + 1], dz = az - s_sh[3 * t + 2];
      d2 = fmin(d2, dx * dx + dy * dy + dz * dz);
    }
    dmin = fmin(dmin, sqrt(d2));
  }
  int own = natm;
  for (int a = 0; a < natm && own == natm; ++a) {
    const double ax = x - s_at[3 * a], ay = y - s_at[3 * a + 1], az = z - s_at[3 * a + 2];
    double d2 = 1e300;
    for (int t = 0; t < 27; ++t) {
      const double dx = ax - s_sh[3 * t], dy = ay - s_sh[3 * t + 1], dz = az - s_sh[3 * t + 2];
      d2 = fmin(d2, dx * dx + dy * dy + dz * dz);
    }
    if (sqrt(d2) <= dmin + tie_atol) own = a;
  }
  owner[g] = own;
}
}  // namespace

extern "C" int isdf_partition_by_atom(isdf_handle h, const double* d_coords, int64_t ngrids, const double* atom_coords, int natm,
                                      const double a[9], double tie_atol, int32_t* d_owner) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, d_coords && atom_coords && a && d_owner && ngrids > 0 && natm > 0 && natm <= 2000 && tie_atol >= 0.0);
  std::vector<double> host(3 * (size_t)natm + 81);
  for (int i = 0; i < 3 * natm; ++i) host[i] = atom_coords[i];
  int t = 0;
  for (int i = -1; i <= 1; ++i)
    for (int j = -1; j <= 1; ++j)
      for (int k = -1; k <= 1; ++k, ++t)
        for (int c = 0; c < 3; ++c) host[3 * natm + 3 * t + c] = i * a[c] + j * a[3 + c] + k * a[6 + c];
  double* d_tab = (double*)isdf_ws(h, "partition_tab", sizeof(double) * host.size());
  if (!d_tab) return ISDF_ERR_HIP;
  HIP_TRY(h, hipMemcpyAsync(d_tab, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  ProfScope ps(h, "partition_kernel[byte]", 28.0 * (double)ngrids);
  hipLaunchKernelGGL(partition_kernel, dim3((unsigned)cdiv(ngrids, 256)), dim3(256), sizeof(double) * host.size(), h->stream,
                     d_coords, ngrids, d_tab, natm, d_tab + 3 * natm, tie_atol, d_owner);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

extern "C" int isdf_select_ip(isdf_handle h, const double* d_ao, int nao, int64_t ld, int nblk,
                              const int64_t* blk_off, const int32_t* nip, double tol,
                              double tie_rtol, double* d_L, int64_t ldL, int64_t* d_piv,
                              int32_t* rank) {
  return isdf_select_ip_cplx(h, d_ao, nao, 0, ld, nblk, blk_off, nip, tol, tie_rtol, d_L, ldL, d_piv, rank);
}

extern "C" int isdf_select_ip_cplx(isdf_handle h, const double* d_ao, int nao, int nh, int64_t ld, int nblk,
                                   const int64_t* blk_off, const int32_t* nip, double tol,
                                   double tie_rtol, double* d_L, int64_t ldL, int64_t* d_piv,
                                   int32_t* rank) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, nh == 0 || 2 * nh == nao);
  ARG_CHECK(h, d_ao && blk_off && nip && d_L && d_piv && rank);
  ARG_CHECK(h, nao > 0 && nblk > 0 && tie_rtol >= 0.0 && tie_rtol < 1.0);
  const int64_t mtot = blk_off[nblk];
  ARG_CHECK(h, blk_off[0] == 0 && mtot > 0 && ld >= mtot && ldL >= mtot);
  int kmax = 0;
  std::vector<int> h_wg_blk;
  std::vector<int64_t> h_wg_lo;
  for (int b = 0; b < nblk; ++b) {
    ARG_CHECK(h, blk_off[b + 1] >= blk_off[b] && nip[b] >= 0);
    if (nip[b] > kmax) kmax = nip[b];
    for (int64_t lo = blk_off[b]; lo < blk_off[b + 1]; lo += TPB) {
      h_wg_blk.push_back(b);
      h_wg_lo.push_back(lo);
    }
  }
  ARG_CHECK(h, kmax > 0);
  const size_t panel_bytes = sizeof(double) * ((size_t)nao * (nh > 0 ? 2 : 1) + kmax);
  const bool lds_panel = panel_bytes <= 64 * 1024;
  const int nwg = (int)h_wg_blk.size();

  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t b_d = al(sizeof(double) * mtot), b_st = al(sizeof(BlkState) * nblk);
  const size_t b_wb = al(sizeof(int) * nwg), b_wl = al(sizeof(int64_t) * nwg);
  const size_t b_off = al(sizeof(int64_t) * (nblk + 1)), b_nip = al(sizeof(int) * nblk);
  const size_t b_pv = al(sizeof(double) * (size_t)nblk * nao) * 2, b_pl = al(sizeof(double) * (size_t)nblk * kmax);
  char* ws = (char*)isdf_ws(h, "select", b_d + b_st + b_wb + b_wl + b_off + b_nip + b_pv + b_pl);
  if (!ws) return ISDF_ERR_HIP;
  double* d_d = (double*)ws; ws += b_d;
  BlkState* d_st = (BlkState*)ws; ws += b_st;
  int* d_wg_blk = (int*)ws; ws += b_wb;
  int64_t* d_wg_lo = (int64_t*)ws; ws += b_wl;
  int64_t* d_off = (int64_t*)ws; ws += b_off;
  int* d_nip = (int*)ws; ws += b_nip;
  double* d_pv = (double*)ws; ws += b_pv;
  double* d_pvr = d_pv + (size_t)nblk * nao;
  double* d_pl = (double*)ws;

  std::vector<BlkState> h_st(nblk);
  for (auto& s : h_st) { s.dmax_bits = 0; s.cand = LLONG_MAX; s.dp = 0; s.tol = 0; s.pivot = -1; s.done = 0; s.rank = 0; }
  HIP_TRY(h, hipMemcpyAsync(d_st, h_st.data(), sizeof(BlkState) * nblk, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_wg_blk, h_wg_blk.data(), sizeof(int) * nwg, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_wg_lo, h_wg_lo.data(), sizeof(int64_t) * nwg, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_off, blk_off, sizeof(int64_t) * (nblk + 1), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_nip, nip, sizeof(int) * nblk, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemsetAsync(d_piv, 0xff, sizeof(int64_t) * (size_t)nblk * kmax, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));   // host staging vectors go out of scope below

  hipLaunchKernelGGL(init_diag_kernel, dim3(nwg), dim3(TPB), 0, h->stream, d_ao, nao, ld, d_wg_blk,
                     d_wg_lo, d_off, d_d, d_st);
  KERNEL_CHECK(h);
  for (int j = 0; j <= kmax; ++j) {
    hipLaunchKernelGGL(find_first_kernel, dim3(nwg), dim3(TPB), 0, h->stream, d_d, d_wg_blk, d_wg_lo,
                       d_off, d_st, tie_rtol);
    hipLaunchKernelGGL(take_pivot_kernel, dim3(nblk), dim3(TPB), 0, h->stream, d_ao, nao, ld, d_L, ldL,
                       d_d, d_off, d_nip, j, tol, kmax, d_st, d_piv, d_pv, d_pl, nh, d_pvr);
    if (j == kmax) break;   // the last take_pivot only marks every block done
    ProfScope ps(h, "select_update_kernel[byte]", 8.0 * (double)mtot * (nao + j + 3));
    if (lds_panel)
      hipLaunchKernelGGL(update_kernel<true>, dim3(nwg), dim3(TPB), sizeof(double) * ((size_t)nao * (nh > 0 ? 2 : 1) + j),
                         h->stream, d_ao, nao, ld, d_L, ldL, d_d, d_wg_blk, d_wg_lo, d_off, j, kmax, d_st, d_pv, d_pl, nh, d_pvr);
    else
      hipLaunchKernelGGL(update_kernel<false>, dim3(nwg), dim3(TPB), 0, h->stream, d_ao, nao, ld, d_L, ldL, d_d,
                         d_wg_blk, d_wg_lo, d_off, j, kmax, d_st, d_pv, d_pl, nh, d_pvr);
  }
  KERNEL_CHECK(h);
  HIP_TRY(h, hipMemcpyAsync(h_st.data(), d_st, sizeof(BlkState) * nblk, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int b = 0; b < nblk; ++b) rank[b] = h_st[b].rank;
  return ISDF_OK;
}

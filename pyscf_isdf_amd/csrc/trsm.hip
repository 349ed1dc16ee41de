// Triangular solves with many right-hand sides, by substitution.
//
//   X (m x n, row-major, ldx)  <-  op(L)^-1 X,    L (m x m) lower triangular, row-major, ldl;  op = L | L^T
//
// An alternative to rocBLAS dtrsm for every place the ISDF fit applies a Cholesky factor (the per-atom block factors D_b
// on the (P, G) fit rows and on the (P, P) matrices, the factor of A' on the (P, P) matrices), selected at run time with
// isdf_set_option(h, "trsm_substitution", 1).  rocBLAS's trsm inverts 128 x 128 diagonal blocks and multiplies; here the
// diagonal blocks (64 rows) are solved by plain forward/backward substitution, one right-hand side per lane, the triangle
// read through the scalar cache, and everything off the diagonal is rocBLAS dgemm; right-sided solves are left-sided ones
// on the transpose.  Measured: the two give the same K to the noise level of the block-Jacobi route on every case tried
// (He2 test cell, diamond 2x2x2 / 4x4x4), rocBLAS is faster (12.4 s against 12.8 s per step at 4x4x4; many small launches
// here), so rocBLAS stays the default; this path is a cross-check that does not share rocBLAS's algorithm.  (Round 1 ran its
// rocprofv3 --pmc passes through it after a profiler crash attributed to rocBLAS's trsm on 1.7M-column right-hand sides; the
// isolated call with those arguments profiles cleanly - profiles/r02_trsm_under_pmc_repro.log - and the default fit route
// never calls dtrsm over the grid, so PMC passes of the default command need no switch.)
#include "common.h"
#include <cstdlib>

namespace {

constexpr int SB = 64;     // rows per substitution block (one register per row and lane)

// rows [0, nb) of X (nb <= 64):  x <- L^-1 x  (TRANS = false)  or  x <- L^-T x  (TRANS = true)
template <bool TRANS>
__global__ __launch_bounds__(256) void subst_kernel(const double* __restrict__ L, int64_t ldl, int nb,
                                                    double* __restrict__ X, int64_t ldx, int64_t n) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  double x[SB];
#pragma unroll
  for (int i = 0; i < SB; ++i) x[i] = (i < nb) ? X[(int64_t)i * ldx + c] : 0.0;
  if (!TRANS) {
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      if (i < nb) {
        double s = x[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s = fma(-L[(int64_t)i * ldl + k], x[k], s);
        x[i] = s / L[(int64_t)i * ldl + i];
      }
    }
  } else {
#pragma unroll
    for (int i = SB - 1; i >= 0; --i) {
      if (i < nb) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < SB; ++k)
          if (k < nb) s = fma(-L[(int64_t)k * ldl + i], x[k], s);
        x[i] = s / L[(int64_t)i * ldl + i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < SB; ++i)
    if (i < nb) X[(int64_t)i * ldx + c] = x[i];
}

// dst (cols x rows, ldd) = src (rows x cols, lds)^T, 32 x 32 tiles through LDS
__global__ void transpose_kernel(const double* __restrict__ src, int64_t lds, int64_t rows, int64_t cols,
                                 double* __restrict__ dst, int64_t ldd) {
  __shared__ double tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;   // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[r * lds + c] : 0.0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int64_t r = c0 + i, c = r0 + tx;        // dst row = src column
    if (r < cols && c < rows) dst[r * ldd + c] = tile[tx][i];
  }
}

// Block-diagonal forward solve over many columns in ONE launch:  X[blk_b, :] <- D_b^-1 X[blk_b, :]  for every block b
// (D_b row-major lower, leading dimension ldd, at D + off_b * ldd + off_b).  One grid column per lane, 64-row sub-blocks
// in registers; the sub-block's coupling to the finished sub-blocks of the same block is applied by re-reading their
// rows (written by this very lane a moment ago: L2), eight at a time, against rows of D_b that every lane shares and
// that therefore come through the scalar cache; then forward substitution with the diagonal tile, as subst_kernel.
// HBM traffic = one read and one write of X (rocBLAS dtrsm block by block moves the same data at about 1 TB/s).
__global__ __launch_bounds__(256) void block_forward_kernel(const double* __restrict__ D, int64_t ldd,
                                                            const int32_t* __restrict__ blk_off,
                                                            double* __restrict__ X, int64_t ldx, int64_t n) {
  const int b = blockIdx.y;
  const int off = blk_off[b], m = blk_off[b + 1] - off;
  if (m <= 0) return;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  const double* Db = D + (int64_t)off * ldd + off;
  double* Xb = X + (int64_t)off * ldx + c;
  const int nsub = (m + 63) >> 6;
  double x[SB];
  for (int j = 0; j < nsub; ++j) {
    const int nbj = min(SB, m - j * SB);
    const double* Lj = Db + (int64_t)(j * SB) * ldd;        // rows of sub-block j
#pragma unroll
    for (int q = 0; q < SB; ++q) x[q] = (q < nbj) ? Xb[(int64_t)(j * SB + q) * ldx] : 0.0;
    for (int r0 = 0; r0 < j * SB; r0 += 8) {               // finished rows of this block, eight at a time
      double v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = Xb[(int64_t)(r0 + t) * ldx];
#pragma unroll
      for (int q = 0; q < SB; ++q) {
        if (q < nbj) {
          const double* Lq = Lj + (int64_t)q * ldd + r0;
#pragma unroll
          for (int t = 0; t < 8; ++t) x[q] = fma(-Lq[t], v[t], x[q]);
        }
      }
    }
    const double* Ld = Lj + j * SB;                          // diagonal tile
#pragma unroll
    for (int q = 0; q < SB; ++q) {
      if (q < nbj) {
        double s_ = x[q];
#pragma unroll
        for (int k = 0; k < q; ++k) s_ = fma(-Ld[(int64_t)q * ldd + k], x[k], s_);
        x[q] = s_ / Ld[(int64_t)q * ldd + q];
      }
    }
#pragma unroll
    for (int q = 0; q < SB; ++q)
      if (q < nbj) Xb[(int64_t)(j * SB + q) * ldx] = x[q];
  }
}

int subst(isdf_handle h, bool trans, const double* L, int64_t ldl, int nb, double* X, int64_t ldx, int64_t n) {
  ProfScope ps(h, trans ? "trsm_subst_kernel<true>[flop]" : "trsm_subst_kernel<false>[flop]", (double)nb * nb * (double)n);
  const dim3 grid((unsigned)cdiv(n, 256));
  if (trans) hipLaunchKernelGGL(subst_kernel<true>, grid, dim3(256), 0, h->stream, L, ldl, nb, X, ldx, n);
  else hipLaunchKernelGGL(subst_kernel<false>, grid, dim3(256), 0, h->stream, L, ldl, nb, X, ldx, n);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

}  // namespace

// X[blk_b, :] <- Dinv_b X[blk_b, :] in place, Dinv = the explicit inverses of the block factors (lower triangular, zeros above
// the diagonal), on the matrix cores.  A workgroup takes one block and TN grid columns: the block's rows of those columns
// are staged in LDS once (they are both the only input and, row tile by row tile, the output), every wave owns row tiles
// of 16 and walks the k range of its tile (lower triangular: k < 16 (tile + 1)) with v_mfma_f64_16x16x4: the A fragment
// is 16 rows x 4 columns of Dinv_b straight from L2 (shared by all column tiles of the block), the B fragments come from
// LDS.  HBM traffic = one read and one write of X; 2 nb^2 n flop per block at MFMA rate instead of nb^2 n scalar-operand
// FMAs (block_forward_kernel above: 0.94 TB/s at configs[2]).
typedef double d4v __attribute__((ext_vector_type(4)));
// k is walked in chunks of 32 with the order inside a chunk PERMUTED consistently for both operands: k-step j of a chunk uses
// k = k0 + 8 fk + j (fk = lane >> 4), so lane (row, fk) needs A[row][k0 + 8 fk + 0..7] - eight CONTIGUOUS doubles, a 64-byte
// run per lane and 256-byte runs per row - instead of eight values strided by four.  The next chunk's eight loads are in
// flight while the current chunk's 8 x TN/16 MFMAs run.
// SQ: the input is squared while it is staged (X holds phi_P phi^T, the pair-gram rows are its element-wise square): saves the
// separate square pass over the rows.
template <int TN, bool SQ>
__global__ __launch_bounds__(256) void block_apply_mfma_kernel(const double* __restrict__ Dinv, int64_t ldd,
                                                               const int32_t* __restrict__ blk_off, double* __restrict__ X,
                                                               int64_t ldx, int64_t n) {
  extern __shared__ double sB[];
  constexpr int LDB = TN + 2;
  const int b = blockIdx.y;
  const int off = blk_off[b], m = blk_off[b + 1] - off;
  if (m <= 0) return;
  const int mpad = (m + 31) & ~31;
  const int64_t c0 = (int64_t)blockIdx.x * TN;
  const int ncol = (int)min((int64_t)TN, n - c0);
  double* Xb = X + (int64_t)off * ldx + c0;
  for (int idx = threadIdx.x; idx < mpad * TN; idx += 256) {
    const int r = idx / TN, c = idx - r * TN;
    double v = (r < m && c < ncol) ? Xb[(int64_t)r * ldx + c] : 0.0;
    if (SQ) v = v * v;
    sB[r * LDB + c] = v;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int fr = lane & 15, fk = lane >> 4;
  for (int rt = wave; rt < (m + 15) / 16; rt += 4) {
    d4v acc[TN / 16];
#pragma unroll
    for (int ct = 0; ct < TN / 16; ++ct) acc[ct] = (d4v){0.0, 0.0, 0.0, 0.0};
    const int arow = rt * 16 + fr;
    const bool rowok = arow < m;
    const double* pa = Dinv + (int64_t)(off + (rowok ? arow : 0)) * ldd + off + 8 * fk;
    const double* pb = sB + (8 * fk) * LDB + fr;
    const int kend = min(mpad, ((rt + 1) * 16 + 31) & ~31);      // lower triangular: k < 16 (rt + 1), in whole chunks
    double a_cur[8], a_nxt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a_cur[j] = (rowok && 8 * fk + j < m) ? pa[j] : 0.0;
    for (int k0 = 0; k0 < kend; k0 += 32) {
      if (k0 + 32 < kend) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a_nxt[j] = (rowok && k0 + 32 + 8 * fk + j < m) ? pa[k0 + 32 + j] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int ct = 0; ct < TN / 16; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[j], pb[(k0 + j) * LDB + ct * 16], acc[ct], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) a_cur[j] = a_nxt[j];
    }
#pragma unroll
    for (int ct = 0; ct < TN / 16; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + fk + 4 * r, col = ct * 16 + fr;
        if (row < m && col < ncol) Xb[(int64_t)row * ldx + col] = acc[ct][r];
      }
  }
}

// Second form of the block apply (round 3): the block inverse lives in REGISTERS and the workgroup walks many column tiles.
// The kernel above re-reads Dinv_b from L2 for every 32-column tile (6.9 M workgroups at configs[2], 100 KB each: more L2
// traffic than the rows' own HBM traffic - it ran at 1.64 TB/s, latency-bound).  Here a workgroup owns ONE block and a
// grid-stride set of 32-column tiles: every wave loads the rows of Dinv_b it needs once (row tiles w and NT-1-w of 16 rows
// each - the lower triangle makes that pair's k range the same for every w: NT/2 + 1 chunks of 32 - as MFMA A fragments, 8
// doubles per chunk and lane, k order permuted inside a chunk as above), then streams: the next tile's rows travel from global
// memory into registers while the current tile's MFMAs run from LDS; one LDS buffer, two barriers per tile.  NCT = column
// sub-tiles of 16 per tile (2: 32 columns = 256-byte row segments), both handled by every wave; waves = NT / 2 pairs.
template <int NT, int NCT, bool SQ>
__global__ __launch_bounds__(64 * ((NT + 1) / 2)) __attribute__((amdgpu_waves_per_eu(1, 2))) void block_apply_reg_kernel(const double* __restrict__ Dinv, int64_t ldd,
                                                                            const int32_t* __restrict__ blk_off,
                                                                            double* __restrict__ X, int64_t ldx, int64_t n) {
  constexpr int TN = 16 * NCT;
  constexpr int LDB = TN + 2;
  constexpr int NPAIR = (NT + 1) / 2;
  constexpr int NTHREADS = 64 * NPAIR;
  constexpr int NCH = NT / 2 + 1;                       // chunks of 32 k per wave (tile w: (w + 2) / 2, tile NT-1-w: (NT + 1 - w) / 2)
  constexpr int MP = 16 * NT;                           // padded rows
  constexpr int NPRE = (MP * TN + NTHREADS - 1) / NTHREADS;
  extern __shared__ double sB[];
  const int b = blockIdx.y;
  const int off = blk_off[b], m = blk_off[b + 1] - off;
  if (m <= 0) return;
  const int pair = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int fr = lane & 15, fk = lane >> 4;
  const int tA = pair, tB = NT - 1 - pair;              // the two row tiles of this wave (tB == tA: the middle tile of an odd NT)
  const int nchA = (tA + 2) / 2, nchB = (tB == tA) ? 0 : (tB + 2) / 2;
  // A fragments: chunk c of the wave's list, 8 contiguous doubles of one row of Dinv_b per lane
  double a[8 * NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const bool isA = c < nchA;
    const int t = isA ? tA : tB;
    const int k0 = 32 * (isA ? c : c - nchA) + 8 * fk;
    const int row = t * 16 + fr;
    const bool live = (c < nchA + nchB) && row < m;
    const double* pa = Dinv + (int64_t)(off + (live ? row : 0)) * ldd + off + k0;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[8 * c + j] = (live && k0 + j < m) ? pa[j] : 0.0;
  }
  const int64_t ntile = (n + TN - 1) / TN;
  double pre[NPRE];
  // this thread's elements of a tile: idx = threadIdx.x + i * NTHREADS -> (row idx / TN, column idx % TN)
  const int r0 = threadIdx.x / TN, cc = threadIdx.x % TN;
  constexpr int RSTEP = NTHREADS / TN;                  // NTHREADS is a multiple of TN (64 | NTHREADS, TN in {16, 32})
  double* const Xrow = X + (int64_t)(off + r0) * ldx + cc;
  auto fetch = [&](int64_t tile) {
    const int64_t c0 = tile * TN;
    const bool colok = c0 + cc < n;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int r = r0 + i * RSTEP;
      pre[i] = (r < m && colok) ? Xrow[(int64_t)i * RSTEP * ldx + c0] : 0.0;
    }
  };
  int64_t tile = blockIdx.x;
  if (tile < ntile) fetch(tile);
  for (; tile < ntile; tile += gridDim.x) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int r = r0 + i * RSTEP;
      if (r < MP) {
        const double v = pre[i];
        sB[r * LDB + cc] = SQ ? v * v : v;
      }
    }
    __syncthreads();
    if (tile + gridDim.x < ntile) fetch(tile + gridDim.x);            // in flight under the MFMAs below
    const int64_t c0 = tile * TN;
    // the column sub-tiles one after the other: one accumulator pair live at a time (both at once cost 16 more VGPRs per
    // sub-tile and spilled for NT >= 12)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      d4v accA = (d4v){0.0, 0.0, 0.0, 0.0}, accB = (d4v){0.0, 0.0, 0.0, 0.0};
      const double* pb = sB + (8 * fk) * LDB + 16 * ct + fr;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c < nchA) {
          const double* q = pb + (32 * c) * LDB;
#pragma unroll
          for (int j = 0; j < 8; ++j) accA = __builtin_amdgcn_mfma_f64_16x16x4f64(a[8 * c + j], q[j * LDB], accA, 0, 0, 0);
        } else if (c < nchA + nchB) {
          const double* q = pb + (32 * (c - nchA)) * LDB;
#pragma unroll
          for (int j = 0; j < 8; ++j) accB = __builtin_amdgcn_mfma_f64_16x16x4f64(a[8 * c + j], q[j * LDB], accB, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);       // keep a chunk's LDS reads with its MFMAs: hoisting all of them costs 100+ VGPRs
      }
      const int64_t col = c0 + ct * 16 + fr;
      if (col < n) {
        double* Xb = X + (int64_t)off * ldx + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rowA = tA * 16 + fk + 4 * r;
          if (rowA < m) Xb[(int64_t)rowA * ldx] = accA[r];
          if (nchB > 0) {
            const int rowB = tB * 16 + fk + 4 * r;
            if (rowB < m) Xb[(int64_t)rowB * ldx] = accB[r];
          }
        }
      }
    }
    __syncthreads();                                                   // every wave is done with the tile in LDS
  }
}

template <int NT, int NCT>
static int block_apply_reg_launch(isdf_handle h, const double* Dinv, int64_t ldd, int nblk, const int32_t* d_off, double* X,
                                  int64_t ldx, int64_t n, bool sq) {
  constexpr int TN = 16 * NCT;
  constexpr int NTHREADS = 64 * ((NT + 1) / 2);
  static_assert(NTHREADS % TN == 0 && (16 * NT * TN) % NTHREADS == 0, "tile elements must divide evenly over the threads");
  const size_t lds = (size_t)16 * NT * (TN + 2) * sizeof(double);
  const int64_t ntile = (n + TN - 1) / TN;
  // workgroups per block: `block_apply_waves` (default 16) times the CU count in all - 2.24 TB/s with 4, 2.6-2.7 with 16-32 at
  // 1.7 M columns (tools/bench_block_apply.py): short serial chains per workgroup matter more than re-reading the block inverse
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ntile, ((int64_t)h->num_cu * h->block_apply_waves + nblk - 1) / nblk));
  if (sq) {
    if (lds > 64 * 1024)
      HIP_TRY(h, hipFuncSetAttribute((const void*)&block_apply_reg_kernel<NT, NCT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(block_apply_reg_kernel<NT, NCT, true>), dim3(gx, (unsigned)nblk), dim3(NTHREADS), lds, h->stream,
                       Dinv, ldd, d_off, X, ldx, n);
  } else {
    if (lds > 64 * 1024)
      HIP_TRY(h, hipFuncSetAttribute((const void*)&block_apply_reg_kernel<NT, NCT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(block_apply_reg_kernel<NT, NCT, false>), dim3(gx, (unsigned)nblk), dim3(NTHREADS), lds, h->stream,
                       Dinv, ldd, d_off, X, ldx, n);
  }
  KERNEL_CHECK(h);
  return ISDF_OK;
}

template <int TN, bool SQ>
static hipError_t block_apply_lds_attr() {
  return hipFuncSetAttribute((const void*)&block_apply_mfma_kernel<TN, SQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

int block_apply_inverse(isdf_handle h, const double* Dinv, int64_t ldd, int nblk, const int32_t* blk_off_host, double* X,
                        int64_t ldx, int64_t n, bool square_input) {
  ARG_CHECK(h, Dinv && X && blk_off_host && nblk > 0 && n > 0 && nblk <= 65535);
  int mmax = 0;
  for (int b = 0; b < nblk; ++b) mmax = std::max(mmax, blk_off_host[b + 1] - blk_off_host[b]);
  const int mpad = (mmax + 31) & ~31;
  ARG_CHECK(h, mmax > 0);
  int32_t* d_off = (int32_t*)isdf_ws(h, "trsm_blk_off", sizeof(int32_t) * (size_t)(nblk + 1));
  if (!d_off) return ISDF_ERR_HIP;
  HIP_TRY(h, hipMemcpyAsync(d_off, blk_off_host, sizeof(int32_t) * (size_t)(nblk + 1), hipMemcpyHostToDevice, h->stream));
  ProfScope ps(h, "block_apply_mfma_kernel[byte]", 16.0 * (double)blk_off_host[nblk] * (double)n);
  // register-resident form for blocks of up to 256 rows (option "block_apply_reg", default on)
  if (h->block_apply_reg && mmax <= 256) {
    const int nt = (mmax + 15) / 16;
    if (nt <= 8) return block_apply_reg_launch<8, 2>(h, Dinv, ldd, nblk, d_off, X, ldx, n, square_input);
    if (nt <= 10) return block_apply_reg_launch<10, 2>(h, Dinv, ldd, nblk, d_off, X, ldx, n, square_input);
    if (nt <= 12) return block_apply_reg_launch<12, 2>(h, Dinv, ldd, nblk, d_off, X, ldx, n, square_input);
    return block_apply_reg_launch<16, 1>(h, Dinv, ldd, nblk, d_off, X, ldx, n, square_input);
  }
  // 64 columns per workgroup while the block's rows fit 52 KB of LDS (three workgroups per CU), else 32, else 16
  // more than 64 KB of dynamic LDS needs the attribute raised once per instantiation (per device: kept in the handle's option map)
#define ISDF_BA_LAUNCH(TN_, SQ_)                                                                                       \
  do {                                                                                                                 \
    if ((size_t)mpad * (TN_ + 2) * sizeof(double) > 64 * 1024) {                                                       \
      HIP_TRY(h, (block_apply_lds_attr<TN_, SQ_>()));                                                                  \
    }                                                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(block_apply_mfma_kernel<TN_, SQ_>), dim3((unsigned)cdiv(n, TN_), (unsigned)nblk),   \
                       dim3(256), (size_t)mpad * (TN_ + 2) * sizeof(double), h->stream, Dinv, ldd, d_off, X, ldx, n);  \
  } while (0)
  if ((size_t)mpad * (64 + 2) * sizeof(double) <= 52 * 1024) {
    if (square_input) ISDF_BA_LAUNCH(64, true); else ISDF_BA_LAUNCH(64, false);
  } else if ((size_t)mpad * (32 + 2) * sizeof(double) <= 80 * 1024) {
    if (square_input) ISDF_BA_LAUNCH(32, true); else ISDF_BA_LAUNCH(32, false);
  } else {
    ARG_CHECK(h, (size_t)mpad * (16 + 2) * sizeof(double) <= 160 * 1024);
    if (square_input) ISDF_BA_LAUNCH(16, true); else ISDF_BA_LAUNCH(16, false);
  }
#undef ISDF_BA_LAUNCH
  KERNEL_CHECK(h);
  return ISDF_OK;
}

int block_forward_solve(isdf_handle h, const double* D, int64_t ldd, int nblk, const int32_t* blk_off_host, double* X,
                        int64_t ldx, int64_t n) {
  ARG_CHECK(h, D && X && blk_off_host && nblk > 0 && n > 0 && nblk <= 65535);
  int32_t* d_off = (int32_t*)isdf_ws(h, "trsm_blk_off", sizeof(int32_t) * (size_t)(nblk + 1));
  if (!d_off) return ISDF_ERR_HIP;
  HIP_TRY(h, hipMemcpyAsync(d_off, blk_off_host, sizeof(int32_t) * (size_t)(nblk + 1), hipMemcpyHostToDevice, h->stream));
  double rows2 = 0.0;
  for (int b = 0; b < nblk; ++b) rows2 += (double)(blk_off_host[b + 1] - blk_off_host[b]) * (blk_off_host[b + 1] - blk_off_host[b]);
  ProfScope ps(h, "block_forward_kernel[byte]", 16.0 * (double)blk_off_host[nblk] * (double)n);
  (void)rows2;
  hipLaunchKernelGGL(block_forward_kernel, dim3((unsigned)cdiv(n, 256), (unsigned)nblk), dim3(256), 0, h->stream, D, ldd, d_off,
                     X, ldx, n);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

int transpose_rm(isdf_handle h, const double* src, int64_t lds, int64_t rows, int64_t cols, double* dst, int64_t ldd) {
  ARG_CHECK(h, src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= rows);
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)), dim3(32, 8), 0, h->stream,
                     src, lds, rows, cols, dst, ldd);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

int trsm_lower_left(isdf_handle h, bool trans, int m, int64_t n, const double* L, int64_t ldl, double* X, int64_t ldx) {
  ARG_CHECK(h, L && X && m > 0 && n > 0 && ldl >= m && ldx >= n);
  // two levels: panels of PB rows are updated with one large dgemm each, the panel's own triangle goes through
  // 64-row substitution blocks with small dgemm updates inside the panel
  const int PB = 512;
  int rc;
  if (!trans) {
    for (int r0 = 0; r0 < m; r0 += PB) {
      const int r1 = std::min(m, r0 + PB);
      if (r0 > 0) {   // X[r0:r1] -= L[r0:r1, 0:r0] X[0:r0]
        rc = gemm_rm(h, 'N', 'N', r1 - r0, n, r0, -1.0, L + (int64_t)r0 * ldl, ldl, X, ldx, 1.0, X + (int64_t)r0 * ldx, ldx);
        if (rc) return rc;
      }
      for (int s0 = r0; s0 < r1; s0 += SB) {
        const int s1 = std::min(r1, s0 + SB);
        if (s0 > r0) {   // X[s0:s1] -= L[s0:s1, r0:s0] X[r0:s0]
          rc = gemm_rm(h, 'N', 'N', s1 - s0, n, s0 - r0, -1.0, L + (int64_t)s0 * ldl + r0, ldl, X + (int64_t)r0 * ldx, ldx,
                       1.0, X + (int64_t)s0 * ldx, ldx);
          if (rc) return rc;
        }
        rc = subst(h, false, L + (int64_t)s0 * ldl + s0, ldl, s1 - s0, X + (int64_t)s0 * ldx, ldx, n);
        if (rc) return rc;
      }
    }
  } else {
    // L^T x = b: from the bottom up; the coupling of rows [a, b) to later rows [b, e) is L[b:e, a:b]^T
    const int npan = (int)cdiv(m, PB);
    for (int p = npan - 1; p >= 0; --p) {
      const int r0 = p * PB, r1 = std::min(m, r0 + PB);
      if (r1 < m) {   // X[r0:r1] -= L[r1:m, r0:r1]^T X[r1:m]
        rc = gemm_rm(h, 'T', 'N', r1 - r0, n, m - r1, -1.0, L + (int64_t)r1 * ldl + r0, ldl, X + (int64_t)r1 * ldx, ldx, 1.0,
                     X + (int64_t)r0 * ldx, ldx);
        if (rc) return rc;
      }
      const int nsb = (int)cdiv(r1 - r0, SB);
      for (int q = nsb - 1; q >= 0; --q) {
        const int s0 = r0 + q * SB, s1 = std::min(r1, s0 + SB);
        if (s1 < r1) {   // X[s0:s1] -= L[s1:r1, s0:s1]^T X[s1:r1]
          rc = gemm_rm(h, 'T', 'N', s1 - s0, n, r1 - s1, -1.0, L + (int64_t)s1 * ldl + s0, ldl, X + (int64_t)s1 * ldx, ldx,
                       1.0, X + (int64_t)s0 * ldx, ldx);
          if (rc) return rc;
        }
        rc = subst(h, true, L + (int64_t)s0 * ldl + s0, ldl, s1 - s0, X + (int64_t)s0 * ldx, ldx, n);
        if (rc) return rc;
      }
    }
  }
  return ISDF_OK;
}

int trsm_lower_right(isdf_handle h, bool trans, int m, int64_t n, const double* L, int64_t ldl, double* X, int64_t ldx) {
  // X (n x m, row-major, ldx) <- X op(L)^-1  ==  ( op(L)^-T X^T )^T
  ARG_CHECK(h, L && X && m > 0 && n > 0 && ldl >= m && ldx >= m);
  double* T = (double*)isdf_ws(h, "trsm_T", sizeof(double) * (size_t)m * (size_t)n);
  if (!T) return ISDF_ERR_HIP;
  int rc = transpose_rm(h, X, ldx, n, m, T, n);
  if (rc) return rc;
  rc = trsm_lower_left(h, !trans, m, n, L, ldl, T, n);
  if (rc) return rc;
  return transpose_rm(h, T, n, m, n, X, ldx);
}

// ---- dispatch ------------------------------------------------------------------------------------------
// Row-major X (m x n) is the column-major X^T (n x m); the row-major lower L is the column-major upper U = L^T:
//   L^-1 X   <->  X^T U^-1      L^-T X  <->  X^T U^-T      (rocBLAS side right)
//   X L^-1   <->  U^-1 X^T      X L^-T  <->  U^-T X^T      (rocBLAS side left)
namespace {
int trsm_block_size() {
  static const int nb = getenv("ISDF_TRSM_NB") ? std::max(64, atoi(getenv("ISDF_TRSM_NB"))) : 1024;
  return nb;
}
// one rocBLAS dtrsm, column-major view (see the table above)
int rb_left(isdf_handle h, bool trans, int m, int64_t n, const double* L, int64_t ldl, double* X, int64_t ldx) {
  const double one = 1.0;
  ProfScope ps(h, "rocblas_dtrsm[flop]", (double)n * m * m);
  BLAS_TRY(h, rocblas_dtrsm(h->blas, rocblas_side_right, rocblas_fill_upper,
                            trans ? rocblas_operation_transpose : rocblas_operation_none, rocblas_diagonal_non_unit,
                            (rocblas_int)n, m, &one, L, (rocblas_int)ldl, X, (rocblas_int)ldx));
  return ISDF_OK;
}
int rb_right(isdf_handle h, bool trans, int m, int64_t n, const double* L, int64_t ldl, double* X, int64_t ldx) {
  const double one = 1.0;
  ProfScope ps(h, "rocblas_dtrsm[flop]", (double)n * m * m);
  BLAS_TRY(h, rocblas_dtrsm(h->blas, rocblas_side_left, rocblas_fill_upper,
                            trans ? rocblas_operation_transpose : rocblas_operation_none, rocblas_diagonal_non_unit, m,
                            (rocblas_int)n, &one, L, (rocblas_int)ldl, X, (rocblas_int)ldx));
  return ISDF_OK;
}
}  // namespace

// Large factors are solved block row by block row (LEFT-looking): block jb first receives the contribution of all
// finished blocks in ONE deep dgemm (the shape rocBLAS runs at the MFMA rate, and it only reads finished rows), then a
// small dtrsm with the diagonal block.  A monolithic rocBLAS trsm reaches 25-43 TF/s on these shapes, this form ~55-65.
int tri_left(isdf_handle h, bool trans, int m, int64_t n, const double* L, int64_t ldl, double* X, int64_t ldx) {
  if (h->trsm_substitution) return trsm_lower_left(h, trans, m, n, L, ldl, X, ldx);
  ARG_CHECK(h, L && X && m > 0 && n > 0 && ldl >= m && ldx >= n && n < 2147483647LL && ldx < 2147483647LL && ldl < 2147483647LL);
  const int NB = trsm_block_size();
  if (m <= NB) return rb_left(h, trans, m, n, L, ldl, X, ldx);
  const int nblk = (int)cdiv(m, NB);
  int rc;
  if (!trans) {
    for (int b = 0; b < nblk; ++b) {
      const int jb = b * NB, nb = std::min(NB, m - jb);
      if (jb > 0) {   // X[jb:jb+nb] -= L[jb:jb+nb, :jb] X[:jb]
        rc = gemm_rm(h, 'N', 'N', nb, n, jb, -1.0, L + (int64_t)jb * ldl, ldl, X, ldx, 1.0, X + (int64_t)jb * ldx, ldx);
        if (rc) return rc;
      }
      rc = rb_left(h, false, nb, n, L + (int64_t)jb * ldl + jb, ldl, X + (int64_t)jb * ldx, ldx);
      if (rc) return rc;
    }
  } else {
    for (int b = nblk - 1; b >= 0; --b) {
      const int jb = b * NB, nb = std::min(NB, m - jb), j1 = jb + nb;
      if (j1 < m) {   // X[jb:j1] -= L[j1:, jb:j1]^T X[j1:]
        rc = gemm_rm(h, 'T', 'N', nb, n, m - j1, -1.0, L + (int64_t)j1 * ldl + jb, ldl, X + (int64_t)j1 * ldx, ldx, 1.0,
                     X + (int64_t)jb * ldx, ldx);
        if (rc) return rc;
      }
      rc = rb_left(h, true, nb, n, L + (int64_t)jb * ldl + jb, ldl, X + (int64_t)jb * ldx, ldx);
      if (rc) return rc;
    }
  }
  return ISDF_OK;
}

// X (n x m) <- X op(L)^-1, block column by block column with the same left-looking idea
int tri_right(isdf_handle h, bool trans, int m, int64_t n, const double* L, int64_t ldl, double* X, int64_t ldx) {
  if (h->trsm_substitution) return trsm_lower_right(h, trans, m, n, L, ldl, X, ldx);
  ARG_CHECK(h, L && X && m > 0 && n > 0 && ldl >= m && ldx >= m && n < 2147483647LL && ldx < 2147483647LL && ldl < 2147483647LL);
  const int NB = trsm_block_size();
  if (m <= NB) return rb_right(h, trans, m, n, L, ldl, X, ldx);
  const int nblk = (int)cdiv(m, NB);
  int rc;
  if (!trans) {
    // X L^-1: columns from the last block to the first; X[:, jb:j1] -= X[:, j1:] L[j1:, jb:j1]
    for (int b = nblk - 1; b >= 0; --b) {
      const int jb = b * NB, nb = std::min(NB, m - jb), j1 = jb + nb;
      if (j1 < m) {
        rc = gemm_rm(h, 'N', 'N', n, nb, m - j1, -1.0, X + j1, ldx, L + (int64_t)j1 * ldl + jb, ldl, 1.0, X + jb, ldx);
        if (rc) return rc;
      }
      rc = rb_right(h, false, nb, n, L + (int64_t)jb * ldl + jb, ldl, X + jb, ldx);
      if (rc) return rc;
    }
  } else {
    // X L^-T: columns from the first block on; X[:, jb:j1] -= X[:, :jb] L[jb:j1, :jb]^T
    for (int b = 0; b < nblk; ++b) {
      const int jb = b * NB, nb = std::min(NB, m - jb);
      if (jb > 0) {
        rc = gemm_rm(h, 'N', 'T', n, nb, jb, -1.0, X, ldx, L + (int64_t)jb * ldl, ldl, 1.0, X + jb, ldx);
        if (rc) return rc;
      }
      rc = rb_right(h, true, nb, n, L + (int64_t)jb * ldl + jb, ldl, X + jb, ldx);
      if (rc) return rc;
    }
  }
  return ISDF_OK;
}

// Dense FP64 products for the ISDF path.
//  * gemm_rm: row-major wrapper over rocBLAS dgemm for the well-shaped products (rocBLAS reaches
//    70-74 TF/s of the 78.6 TF/s FP64 MFMA peak on them: profiles/r01_probe_rocblas_hipfft_mfma64.log).
//  * gemm_nt_f64: C = alpha * A * (B .* kscale)^T + beta * C with K contiguous in BOTH operands and
//    K >> M, N — the shape of W = V Theta^T (K = ngrids) and of vj = ao (v .* ao)^T.  rocBLAS runs
//    this shape at 13-34 TF/s (profiles/r01_gemm_variants.log), so it gets hand-written v_mfma_f64_16x16x4_f64
//    kernels - three variants share the work-unit scheme below (the launcher picks: B when M fills 256-row tiles,
//    D for other aligned operands, A for unaligned ones; 69 TF/s = 0.88 of peak on the W product):
//      - A: 128x128 output tile per 256-thread workgroup, each wave a 64x64 sub-tile = 4x4 MFMA
//        accumulators (128 VGPRs), both operands through LDS;  B: 256x128, 8 waves, one workgroup per CU;
//        D: 128x128, A operand straight from global memory into MFMA registers, only B through LDS;
//      - K is cut into slabs; one work unit = (slab, tile).  Units are ordered slab-major with the
//        row tile fastest so that workgroups that share an operand panel run at the same time, and
//        the block id is remapped so that such neighbours land on the same XCD (its L2);
//      - operands are staged global -> registers -> LDS in 16-deep K chunks (one full 128-B line per
//        row per chunk), double buffered, rows padded to 136 B so that the ds_read_b64 fragment
//        reads are bank-conflict free;
//      - partial tiles of the slabs go to a workspace and are summed in a fixed order (deterministic,
//        no float atomics);
//      - B and D: the instruction ORDER inside a chunk is pinned with sched_group_barrier - one LDS read (of the next
//        k-step), LDS write (of the next chunk) or global load (of the chunk after next) after every second MFMA, the
//        chunk's single barrier after k-step 2, the next chunk's first fragments prefetched under k-step 3.  Worth
//        64 -> 69 TF/s, and it keeps co-scheduled workgroups in step so that shared operand panels hit in L2 (fabric
//        traffic 3.9x -> 1.3x the algorithmic bytes).
//    Algorithmic work: 2*M*N*K flop; bound: FP64 MFMA (78.6 TF/s).
#include "common.h"
#include <cstdlib>

int gemm_rm(isdf_handle h, char opA, char opB, int64_t M, int64_t N, int64_t K, double alpha,
            const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
            int64_t ldc) {
  // Row-major C = op(A) op(B)  <=>  column-major C^T = op(B)^T op(A)^T.
  ARG_CHECK(h, M < 2147483647LL && N < 2147483647LL && K < 2147483647LL && lda < 2147483647LL &&
                   ldb < 2147483647LL && ldc < 2147483647LL);
  const rocblas_operation ta = (opA == 'N') ? rocblas_operation_none : rocblas_operation_transpose;
  const rocblas_operation tb = (opB == 'N') ? rocblas_operation_none : rocblas_operation_transpose;
  ProfScope ps(h, "rocblas_dgemm[flop]", 2.0 * M * N * (double)K);
  BLAS_TRY(h, rocblas_dgemm(h->blas, tb, ta, (rocblas_int)N, (rocblas_int)M, (rocblas_int)K, &alpha, B,
                            (rocblas_int)ldb, A, (rocblas_int)lda, &beta, C, (rocblas_int)ldc));
  return ISDF_OK;
}

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDT = 17;          // doubles per LDS row: 16 + 1 pad (136 B = 34 banks): the fragment reads (ds_read2_b64,
                                 // banked mod 32 in 16-lane groups) hit 16 distinct bank pairs
constexpr int TPB = 256;

struct GemmArgs {
  const double* A; int64_t lda;
  const double* B; int64_t ldb;
  const double* kscale;          // optional per-k scale applied to B (nullptr: none)
  double* P;                     // partials [nslab][M][N] (or C itself when nslab == 1 && direct)
  int64_t ldp, slab_stride;
  int M, N;
  int64_t K, kslab;              // kslab multiple of BK
  int ntm, ntn, nslab;
  int64_t nunits, nunits_pad;    // nunits_pad: rounded up to a multiple of 8 (XCD remap)
  double alpha, beta;            // used only when direct
  int direct;
};

// FAST: lda, ldb even (16-byte aligned rows given 16-byte aligned bases) and no K tail handling
// needed inside a chunk (K % BK == 0).  Otherwise the generic path loads element-wise with guards.
template <bool FAST>
__global__ __launch_bounds__(TPB, 2) void gemm_nt_mfma_kernel(GemmArgs g) {
  __shared__ double sA[2][BM * LDT];
  __shared__ double sB[2][BN * LDT];

  // XCD-aware unit id: hardware deals consecutive block ids round-robin over 8 XCDs; give each XCD a
  // contiguous range of units so that neighbours (which share operand panels) share an L2.
  const int64_t bid = blockIdx.x;
  const int64_t per_xcd = g.nunits_pad / 8;
  const int64_t unit = (bid % 8) * per_xcd + bid / 8;
  if (unit >= g.nunits) return;
  const int tm = (int)(unit % g.ntm);
  const int tn = (int)((unit / g.ntm) % g.ntn);
  const int slab = (int)(unit / ((int64_t)g.ntm * g.ntn));
  const int64_t k0 = (int64_t)slab * g.kslab;
  const int64_t k1 = (k0 + g.kslab < g.K) ? k0 + g.kslab : g.K;
  const int nchunks = (int)((k1 - k0 + BK - 1) / BK);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // staging map: piece p = tid + 256*i, i < 4: row = p / 8, 16-byte segment = p % 8
  const int srow = tid >> 3;            // 0..31, +32*i
  const int sseg = tid & 7;             // k offset sseg*2
  // row offsets are recomputed per chunk (cheap next to 64 MFMAs) instead of keeping 8 pointers live;
  // out-of-range rows are clamped: they are computed but never stored
  const int arow0 = tm * BM + srow, brow0 = tn * BN + srow;
  const int mlast = g.M - 1, nlast = g.N - 1;
  const double* __restrict__ Ag = g.A;
  const double* __restrict__ Bg = g.B;
  const int64_t lda = g.lda, ldb = g.ldb;

  // staging registers as named scalars (arrays captured by the helpers below ended up in scratch)
  double2 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define ISDF_LOAD_ONE(I, RA, RB)                                                              \
  {                                                                                           \
    const int ra = min(arow0 + 32 * I, mlast), rb = min(brow0 + 32 * I, nlast);               \
    const double* pa = Ag + (int64_t)ra * lda + k;                                            \
    const double* pb = Bg + (int64_t)rb * ldb + k;                                            \
    if (FAST) {                                                                               \
      RA = *reinterpret_cast<const double2*>(pa);                                             \
      RB = *reinterpret_cast<const double2*>(pb);                                             \
    } else {                                                                                  \
      RA.x = v0 ? pa[0] : 0.0; RA.y = v1 ? pa[1] : 0.0;                                       \
      RB.x = v0 ? pb[0] : 0.0; RB.y = v1 ? pb[1] : 0.0;                                       \
    }                                                                                         \
    RB.x *= s0; RB.y *= s1;                                                                   \
  }
#define ISDF_LOAD_CHUNK(C)                                                                    \
  {                                                                                           \
    const int64_t k = k0 + (int64_t)(C) * BK + sseg * 2;                                      \
    const bool v0 = FAST || (k < k1), v1 = FAST || (k + 1 < k1);                              \
    double s0 = 1.0, s1 = 1.0;                                                                \
    if (g.kscale) { if (v0) s0 = g.kscale[k]; if (v1) s1 = g.kscale[k + 1]; }                 \
    ISDF_LOAD_ONE(0, ra0, rb0) ISDF_LOAD_ONE(1, ra1, rb1)                                     \
    ISDF_LOAD_ONE(2, ra2, rb2) ISDF_LOAD_ONE(3, ra3, rb3)                                     \
  }
#define ISDF_STORE_ONE(BUF, I, RA, RB)                                                        \
  { double* qa = &sA[BUF][(srow + 32 * I) * LDT + sseg * 2];                                  \
    double* qb = &sB[BUF][(srow + 32 * I) * LDT + sseg * 2];                                  \
    qa[0] = RA.x; qa[1] = RA.y; qb[0] = RB.x; qb[1] = RB.y; }
#define ISDF_STORE_CHUNK(BUF)                                                                 \
  { ISDF_STORE_ONE(BUF, 0, ra0, rb0) ISDF_STORE_ONE(BUF, 1, ra1, rb1)                         \
    ISDF_STORE_ONE(BUF, 2, ra2, rb2) ISDF_STORE_ONE(BUF, 3, ra3, rb3) }

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  const int frow = lane & 15, fk = lane >> 4;
  ISDF_LOAD_CHUNK(0)
  ISDF_STORE_CHUNK(0)
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) ISDF_LOAD_CHUNK(c + 1)
    const double* pa = &sA[buf][(wm * 64 + frow) * LDT + fk];
    const double* pb = &sB[buf][(wn * 64 + frow) * LDT + fk];
    // fragments of k-step kk+1 are fetched from LDS while the 16 MFMAs of k-step kk run
    double a0[4], b0[4], a1[4], b1[4];
#define ISDF_FRAGS(KK, AF, BF)                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
      AF[i] = pa[i * 16 * LDT + (KK) * 4];                                                    \
      BF[i] = pb[i * 16 * LDT + (KK) * 4];                                                    \
    }
#define ISDF_MFMA16(AF, BF)                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(AF[i], BF[j], acc[i][j], 0, 0, 0);
    ISDF_FRAGS(0, a0, b0)
    ISDF_FRAGS(1, a1, b1)
    ISDF_MFMA16(a0, b0)
    ISDF_FRAGS(2, a0, b0)
    ISDF_MFMA16(a1, b1)
    ISDF_FRAGS(3, a1, b1)
    ISDF_MFMA16(a0, b0)
    ISDF_MFMA16(a1, b1)
    if (c + 1 < nchunks) ISDF_STORE_CHUNK(buf ^ 1)
    __syncthreads();
  }

  // epilogue: D[row = (lane>>4) + 4r][col = lane&15] per 16x16 accumulator
  double* out = g.P + (int64_t)slab * g.slab_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * BM + wm * 64 + i * 16 + (lane >> 4) + 4 * r;
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = tn * BN + wn * 64 + j * 16 + (lane & 15);
        if (col >= g.N) continue;
        double* q = out + (int64_t)row * g.ldp + col;
        const double v = acc[i][j][r];
        if (g.direct) *q = (g.beta == 0.0) ? g.alpha * v : g.alpha * v + g.beta * (*q);
        else *q = v;
      }
    }
  }
}

// ---- variant B: 256x128 tile, 8 waves (one workgroup per CU), operand loads two chunks ahead ------
// Motivation (profiles/r01_pmc_gemm_nt_mfma_v1.json): with 128x128 tiles every workgroup streams its
// operands from the fabric (sharers run in lockstep and all miss in L2), and the single register
// staging set gives a load only one chunk of MFMAs (~3.5 us) to land: ~14% of the time is spent in
// s_waitcnt vmcnt(0).  Here the B panel is shared by twice as many rows inside the workgroup, and two
// staging register sets keep TWO chunks of loads in flight.
constexpr int BM2 = 256;
constexpr int TPB2 = 512;

template <bool SCALED>
__global__ __launch_bounds__(TPB2, 2) void gemm_nt_mfma_kernel_b(GemmArgs g) {
  // aligned path only: K % 32 == 0, kslab % 32 == 0 (an even number of 16-deep chunks per slab),
  // 16-byte aligned rows.  No conditionals in the main loop: the tail re-loads the last chunk.
  extern __shared__ double smem[];                      // [2][BM2*LDT] A | [2][BN*LDT] B
  double* sA = smem;
  double* sB = smem + 2 * BM2 * LDT;
  const int64_t bid = blockIdx.x;
  const int64_t per_xcd = g.nunits_pad / 8;
  const int64_t unit = (bid % 8) * per_xcd + bid / 8;
  if (unit >= g.nunits) return;
  const int tm = (int)(unit % g.ntm);
  const int tn = (int)((unit / g.ntm) % g.ntn);
  const int slab = (int)(unit / ((int64_t)g.ntm * g.ntn));
  const int64_t k0 = (int64_t)slab * g.kslab;
  const int64_t k1 = (k0 + g.kslab < g.K) ? k0 + g.kslab : g.K;
  const int nchunks = (int)((k1 - k0) / BK);            // even
  const int last = nchunks - 1;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;              // 4 x 2 waves, 64x64 each
  const int srow = tid >> 3;                            // 0..63
  const int sseg = tid & 7;
  const int mlast = g.M - 1, nlast = g.N - 1;
  // six row pointers (A: rows srow + 64 i, B: rows srow + 64 i), clamped at the matrix edge
  const double* pA0 = g.A + (int64_t)min(tm * BM2 + srow, mlast) * g.lda + k0 + sseg * 2;
  const double* pA1 = g.A + (int64_t)min(tm * BM2 + srow + 64, mlast) * g.lda + k0 + sseg * 2;
  const double* pA2 = g.A + (int64_t)min(tm * BM2 + srow + 128, mlast) * g.lda + k0 + sseg * 2;
  const double* pA3 = g.A + (int64_t)min(tm * BM2 + srow + 192, mlast) * g.lda + k0 + sseg * 2;
  const double* pB0 = g.B + (int64_t)min(tn * BN + srow, nlast) * g.ldb + k0 + sseg * 2;
  const double* pB1 = g.B + (int64_t)min(tn * BN + srow + 64, nlast) * g.ldb + k0 + sseg * 2;
  const double* pS = SCALED ? g.kscale + k0 + sseg * 2 : nullptr;

  double2 xa0, xa1, xa2, xa3, xb0, xb1, ya0, ya1, ya2, ya3, yb0, yb1;
#define ISDF_LOADB(C, A0, A1, A2, A3, B0, B1)                                                 \
  {                                                                                           \
    const int off = (C) * BK;                                                                 \
    A0 = *reinterpret_cast<const double2*>(pA0 + off);                                        \
    A1 = *reinterpret_cast<const double2*>(pA1 + off);                                        \
    A2 = *reinterpret_cast<const double2*>(pA2 + off);                                        \
    A3 = *reinterpret_cast<const double2*>(pA3 + off);                                        \
    B0 = *reinterpret_cast<const double2*>(pB0 + off);                                        \
    B1 = *reinterpret_cast<const double2*>(pB1 + off);                                        \
    if (SCALED) {                                                                             \
      const double2 sc = *reinterpret_cast<const double2*>(pS + off);                         \
      B0.x *= sc.x; B0.y *= sc.y; B1.x *= sc.x; B1.y *= sc.y;                                 \
    }                                                                                         \
  }
#define ISDF_ST1(P, R) { double* q_ = (P); q_[0] = R.x; q_[1] = R.y; }
#define ISDF_STOREB(BUF, A0, A1, A2, A3, B0, B1)                                              \
  {                                                                                           \
    double* qa = sA + (BUF) * BM2 * LDT + srow * LDT + sseg * 2;                              \
    double* qb = sB + (BUF) * BN * LDT + srow * LDT + sseg * 2;                               \
    ISDF_ST1(qa, A0) ISDF_ST1(qa + 64 * LDT, A1) ISDF_ST1(qa + 128 * LDT, A2)                 \
    ISDF_ST1(qa + 192 * LDT, A3) ISDF_ST1(qb, B0) ISDF_ST1(qb + 64 * LDT, B1)                 \
  }

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  const int frow = lane & 15, fk = lane >> 4;

#define ISDF_FRAGB(KK, AF, BF)                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
      AF[i] = pa[i * 16 * LDT + (KK) * 4];                                                    \
      BF[i] = pb[i * 16 * LDT + (KK) * 4];                                                    \
    }
#define ISDF_MFMAB(AF, BF)                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(AF[i], BF[j], acc[i][j], 0, 0, 0);
  // issue order inside a k-step region: one LDS fragment read (of the NEXT k-step) after every second MFMA, so that the
  // reads trickle in under the MFMAs instead of in one burst (+1%); the LDS writes of the next chunk are spread the same
  // way over the last k-step, which leaves only the barrier at the end of the chunk
#define ISDF_INTERLEAVE() _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#define ISDF_INTERLEAVE_W() _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
  // first k-step: the six global loads of the chunk after next ride along (one per MFMA pair, after the fragment read)
#define ISDF_INTERLEAVE_L() _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                \
    if (q_ < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
  // exactly two fragment sets live (the scheduler would otherwise hoist all four k-steps' reads and
  // spill): reads of k-step kk+1 are issued before the MFMAs of kk, fenced by sched_barrier
  // One chunk = four k-steps.  Fragment sets alternate (k0: set 0, k1: set 1, k2: set 0, k3: set 1); each region issues
  // the LDS reads of the NEXT k-step under its own MFMAs.  The chunk's single barrier sits after k-step 2: by then every
  // wave has issued and completed (s_waitcnt in __syncthreads) all its reads of the current buffer and has written its
  // part of the next one, so k-step 3 can already prefetch the first fragments of the next chunk from the other buffer
  // and the MFMA stream runs across the chunk boundary without the post-barrier bubble.  A wave can only reach the next
  // chunk's writes of this buffer after passing this barrier, i.e. after all waves finished reading it.
  const int aoff = (wm * 64 + frow) * LDT + fk, boff = (wn * 64 + frow) * LDT + fk;
  double a0[4], b0[4], a1[4], b1[4];
#define ISDF_FRAGQ(BUF, KK, AF, BF)                                                           \
  {                                                                                           \
    const double* pa = sA + (BUF) * BM2 * LDT + aoff;                                         \
    const double* pb = sB + (BUF) * BN * LDT + boff;                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
      AF[i] = pa[i * 16 * LDT + (KK) * 4];                                                    \
      BF[i] = pb[i * 16 * LDT + (KK) * 4];                                                    \
    }                                                                                         \
  }
#define ISDF_INTERLEAVE_RW() _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {               \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                \
    if (q_ < 6) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#define ISDF_CHUNKB(BUF, LOAD_AHEAD, STORE_NEXT)                                              \
  {                                                                                           \
    ISDF_FRAGQ(BUF, 1, a1, b1)                                                                \
    LOAD_AHEAD                                                                                \
    ISDF_MFMAB(a0, b0)                                                                        \
    ISDF_INTERLEAVE_L()                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    ISDF_FRAGQ(BUF, 2, a0, b0)                                                                \
    ISDF_MFMAB(a1, b1)                                                                        \
    ISDF_INTERLEAVE()                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    ISDF_FRAGQ(BUF, 3, a1, b1)                                                                \
    STORE_NEXT                                                                                \
    ISDF_MFMAB(a0, b0)                                                                        \
    ISDF_INTERLEAVE_RW()                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    __syncthreads();                                                                          \
    ISDF_FRAGQ(1 - (BUF), 0, a0, b0)                                                          \
    ISDF_MFMAB(a1, b1)                                                                        \
    ISDF_INTERLEAVE()                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  }

  // prologue: chunk 0 -> LDS buffer 0; chunk 1 in flight in set Y; first fragments of chunk 0
  ISDF_LOADB(0, xa0, xa1, xa2, xa3, xb0, xb1)
  ISDF_LOADB(1, ya0, ya1, ya2, ya3, yb0, yb1)
  ISDF_STOREB(0, xa0, xa1, xa2, xa3, xb0, xb1)
  __syncthreads();
  ISDF_FRAGQ(0, 0, a0, b0)
  __builtin_amdgcn_sched_barrier(0);
  // steady state, two chunks per iteration: while chunk c computes, chunk c+1 sits in registers and
  // chunk c+2 is being loaded, so every load has two chunks of MFMAs to land
  for (int c = 0; c < nchunks; c += 2) {
    ISDF_CHUNKB(0, ISDF_LOADB(min(c + 2, last), xa0, xa1, xa2, xa3, xb0, xb1), ISDF_STOREB(1, ya0, ya1, ya2, ya3, yb0, yb1))
    ISDF_CHUNKB(1, ISDF_LOADB(min(c + 3, last), ya0, ya1, ya2, ya3, yb0, yb1), ISDF_STOREB(0, xa0, xa1, xa2, xa3, xb0, xb1))
  }

  double* out = g.P + (int64_t)slab * g.slab_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * BM2 + wm * 64 + i * 16 + (lane >> 4) + 4 * r;
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = tn * BN + wn * 64 + j * 16 + (lane & 15);
        if (col >= g.N) continue;
        double* q = out + (int64_t)row * g.ldp + col;
        const double v = acc[i][j][r];
        if (g.direct) *q = (g.beta == 0.0) ? g.alpha * v : g.alpha * v + g.beta * (*q);
        else *q = v;
      }
    }
  }
}

// ---- variant D: A straight from global memory into MFMA operand registers, only B through LDS ----------------------
// 128 x 128 tile, four waves stacked along M (each 32 rows x all 128 columns: 2 x 8 accumulators), two workgroups per
// CU.  The rows of A are private to a wave, so staging them through LDS buys nothing: lane (r = lane & 15, kq = lane >> 4)
// loads A[row r][16 c + 4 kq .. + 3] (32 contiguous bytes; the four kq lanes of a row cover one 128-byte line) and uses
// element j in the j-th MFMA of the chunk.  That permutes the order of the k index inside a 16-deep chunk
// (k = 4 kq + j instead of 4 j + kq) - the B fragments are read from LDS with the same map, the sum is the same set of
// products.  Only B (128 rows x 16) is written to LDS: a third of variant B's LDS writes per flop, 35 KB of LDS per
// workgroup, and the two resident workgroups hide each other's barrier.
constexpr int TPBD = 256;

template <bool SCALED>
__global__ __launch_bounds__(TPBD, 2) void gemm_nt_mfma_kernel_d(GemmArgs g) {
  __shared__ double sB[2][BN * LDT];
  const int64_t bid = blockIdx.x;
  const int64_t per_xcd = g.nunits_pad / 8;
  const int64_t unit = (bid % 8) * per_xcd + bid / 8;
  if (unit >= g.nunits) return;
  const int tm = (int)(unit % g.ntm);
  const int tn = (int)((unit / g.ntm) % g.ntn);
  const int slab = (int)(unit / ((int64_t)g.ntm * g.ntn));
  const int64_t k0 = (int64_t)slab * g.kslab;
  const int64_t k1 = (k0 + g.kslab < g.K) ? k0 + g.kslab : g.K;
  const int nchunks = (int)((k1 - k0) / BK);            // even (kslab is a multiple of 2 BK, K % 32 == 0)
  const int last = nchunks - 1;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fkq = lane >> 4;
  const int mlast = g.M - 1, nlast = g.N - 1;
  // A: two 16-row groups of this wave's 32 rows; four consecutive k per lane
  const double* pA0 = g.A + (int64_t)min(tm * BM + wave * 32 + frow, mlast) * g.lda + k0 + 4 * fkq;
  const double* pA1 = g.A + (int64_t)min(tm * BM + wave * 32 + 16 + frow, mlast) * g.lda + k0 + 4 * fkq;
  // B staging: 128 rows x 16 doubles per chunk = 8 doubles per thread: rows srow and srow + 64, k offset 4 sseg .. + 3
  const int srow = tid >> 2, sseg = tid & 3;
  const double* pB0 = g.B + (int64_t)min(tn * BN + srow, nlast) * g.ldb + k0 + 4 * sseg;
  const double* pB1 = g.B + (int64_t)min(tn * BN + srow + 64, nlast) * g.ldb + k0 + 4 * sseg;
  const double* pS = SCALED ? g.kscale + k0 + 4 * sseg : nullptr;

  // register sets: X / Y alternate between chunks (A operands + B staging)
  double2 xa00, xa01, xa10, xa11, xb00, xb01, xb10, xb11;
  double2 ya00, ya01, ya10, ya11, yb00, yb01, yb10, yb11;
#define ISDF_LOADD(C, A00, A01, A10, A11, B00, B01, B10, B11)                                 \
  {                                                                                           \
    const int off = (C) * BK;                                                                 \
    A00 = *reinterpret_cast<const double2*>(pA0 + off);                                       \
    A01 = *reinterpret_cast<const double2*>(pA0 + off + 2);                                   \
    A10 = *reinterpret_cast<const double2*>(pA1 + off);                                       \
    A11 = *reinterpret_cast<const double2*>(pA1 + off + 2);                                   \
    B00 = *reinterpret_cast<const double2*>(pB0 + off);                                       \
    B01 = *reinterpret_cast<const double2*>(pB0 + off + 2);                                   \
    B10 = *reinterpret_cast<const double2*>(pB1 + off);                                       \
    B11 = *reinterpret_cast<const double2*>(pB1 + off + 2);                                   \
    if (SCALED) {                                                                             \
      const double2 s0 = *reinterpret_cast<const double2*>(pS + off);                         \
      const double2 s1 = *reinterpret_cast<const double2*>(pS + off + 2);                     \
      B00.x *= s0.x; B00.y *= s0.y; B01.x *= s1.x; B01.y *= s1.y;                             \
      B10.x *= s0.x; B10.y *= s0.y; B11.x *= s1.x; B11.y *= s1.y;                             \
    }                                                                                         \
  }
#define ISDF_STORED(BUF, B00, B01, B10, B11)                                                  \
  {                                                                                           \
    double* q0 = &sB[BUF][srow * LDT + 4 * sseg];                                             \
    double* q1 = &sB[BUF][(srow + 64) * LDT + 4 * sseg];                                      \
    q0[0] = B00.x; q0[1] = B00.y; q0[2] = B01.x; q0[3] = B01.y;                               \
    q1[0] = B10.x; q1[1] = B10.y; q1[2] = B11.x; q1[3] = B11.y;                               \
  }

  d4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  // Same issue order as variant B: the B fragments of the NEXT k-step are read from LDS under the MFMAs of the current one
  // (two fragment sets), the chunk's single barrier sits after k-step 2, and k-step 3 already prefetches the first
  // fragments of the next chunk from the other buffer.
  double bf0[8], bf1[8];
#define ISDF_FRAGD(BUF, J, BF)                                                                \
  {                                                                                           \
    const double* pb = &sB[BUF][frow * LDT + 4 * fkq + (J)];                                  \
    _Pragma("unroll") for (int jn = 0; jn < 8; ++jn) BF[jn] = pb[jn * 16 * LDT];              \
  }
#define ISDF_MFMAD(AV0, AV1, BF)                                                              \
    _Pragma("unroll") for (int jn = 0; jn < 8; ++jn) {                                        \
      acc[0][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(AV0, BF[jn], acc[0][jn], 0, 0, 0);    \
      acc[1][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(AV1, BF[jn], acc[1][jn], 0, 0, 0);    \
    }
#define ISDF_ILD_R() _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#define ISDF_ILD_RL() _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#define ISDF_ILD_RW() _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                \
    if (q_ < 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#define ISDF_CHUNKD(BUF, A00, A01, A10, A11, LOAD_AHEAD, STORE_NEXT)                          \
  {                                                                                           \
    ISDF_FRAGD(BUF, 1, bf1)                                                                   \
    ISDF_MFMAD(A00.x, A10.x, bf0)                                                             \
    ISDF_ILD_R()                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    ISDF_FRAGD(BUF, 2, bf0)                                                                   \
    ISDF_MFMAD(A00.y, A10.y, bf1)                                                             \
    ISDF_ILD_R()                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    ISDF_FRAGD(BUF, 3, bf1)                                                                   \
    STORE_NEXT                                                                                \
    ISDF_MFMAD(A01.x, A11.x, bf0)                                                             \
    ISDF_ILD_RW()                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    __syncthreads();                                                                          \
    ISDF_FRAGD(1 - (BUF), 0, bf0)                                                             \
    ISDF_MFMAD(A01.y, A11.y, bf1)                                                             \
    LOAD_AHEAD                                                                                \
    ISDF_ILD_RL()                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  }

  // prologue: chunk 0 in set X (its B part published in buffer 0), chunk 1 in flight in set Y, first fragments of chunk 0
  ISDF_LOADD(0, xa00, xa01, xa10, xa11, xb00, xb01, xb10, xb11)
  ISDF_LOADD(min(1, last), ya00, ya01, ya10, ya11, yb00, yb01, yb10, yb11)
  ISDF_STORED(0, xb00, xb01, xb10, xb11)
  __syncthreads();
  ISDF_FRAGD(0, 0, bf0)
  __builtin_amdgcn_sched_barrier(0);
  for (int c = 0; c < nchunks; c += 2) {
    // chunk c: A operands in X, B in buffer 0, chunk c+1 sits in Y (its B part goes to buffer 1 during k-step 2);
    // X is refilled with chunk c+2 once its last use (k-step 3) has been issued
    ISDF_CHUNKD(0, xa00, xa01, xa10, xa11, ISDF_LOADD(min(c + 2, last), xa00, xa01, xa10, xa11, xb00, xb01, xb10, xb11),
                ISDF_STORED(1, yb00, yb01, yb10, yb11))
    ISDF_CHUNKD(1, ya00, ya01, ya10, ya11, ISDF_LOADD(min(c + 3, last), ya00, ya01, ya10, ya11, yb00, yb01, yb10, yb11),
                ISDF_STORED(0, xb00, xb01, xb10, xb11))
  }

  double* out = g.P + (int64_t)slab * g.slab_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * BM + wave * 32 + i * 16 + (lane >> 4) + 4 * r;
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = tn * BN + j * 16 + (lane & 15);
        if (col >= g.N) continue;
        double* q = out + (int64_t)row * g.ldp + col;
        const double v = acc[i][j][r];
        if (g.direct) *q = (g.beta == 0.0) ? g.alpha * v : g.alpha * v + g.beta * (*q);
        else *q = v;
      }
    }
  }
}

// ---- NN form: C = A B (optionally squared element-wise), A (M x K) with K contiguous, B (K x N) with N contiguous, K short
// (the AO count) and N long (the grid): the pair-density rows (phi_P^T phi)^2 of the fit.  Same 256x128 tile, 8 waves, register
// double buffering and pinned issue order as variant B; what changes is the B operand: a chunk is 16 full 1-KB rows of the
// K x N matrix (one row per wave and load: perfectly coalesced), written to LDS as it comes ([k][n], rows padded to 144 doubles
// so that the four k-groups of a fragment read start 32 banks apart) and read back as fragments B[k = 4 kk + fk][n] - 16
// contiguous doubles per k-group.  One unit = one output tile over the whole K (no slabs: K is ~1e3, M N is ~1e10); units are
// dealt to the XCDs in super-tiles of stm x stn tiles (32 = the workgroups one XCD runs at a time) so that the co-running
// workgroups share stm A panels and stn B panels in that XCD's L2.
constexpr int LDN = 144;

struct GemmNNArgs {
  const double* A; int64_t lda;
  const double* B; int64_t ldb;
  double* C; int64_t ldc;
  int M, N, K;                   // K % 32 == 0, N even
  int ntm, ntn, stm, stn, ngm;   // tiles, super-tile shape, super-tiles along M
  int64_t nunits, nunits_pad;
};

template <bool SQ>
__global__ __launch_bounds__(TPB2, 2) void gemm_nn_mfma_kernel(GemmNNArgs g) {
  extern __shared__ double smem[];                      // [2][BM2*LDT] A | [2][BK*LDN] B
  double* sA = smem;
  double* sB = smem + 2 * BM2 * LDT;
  const int64_t bid = blockIdx.x;
  const int64_t per_xcd = g.nunits_pad / 8;
  const int64_t unit = (bid % 8) * per_xcd + bid / 8;
  if (unit >= g.nunits) return;
  const int per = g.stm * g.stn;
  const int64_t grp = unit / per;
  const int within = (int)(unit - grp * per);
  const int tm = (int)(grp % g.ngm) * g.stm + within % g.stm;
  const int tn = (int)(grp / g.ngm) * g.stn + within / g.stm;
  if (tm >= g.ntm || tn >= g.ntn) return;
  const int nchunks = g.K / BK;                         // even
  const int last = nchunks - 1;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;              // 4 x 2 waves, 64x64 each
  const int srow = tid >> 3, sseg = tid & 7;            // A staging: row srow + 64 i, k offset 2 sseg
  const int krow = tid >> 6, nseg = tid & 63;           // B staging: k rows krow and krow + 8, columns 2 nseg, 2 nseg + 1
  const int mlast = g.M - 1;
  const double* pA0 = g.A + (int64_t)min(tm * BM2 + srow, mlast) * g.lda + sseg * 2;
  const double* pA1 = g.A + (int64_t)min(tm * BM2 + srow + 64, mlast) * g.lda + sseg * 2;
  const double* pA2 = g.A + (int64_t)min(tm * BM2 + srow + 128, mlast) * g.lda + sseg * 2;
  const double* pA3 = g.A + (int64_t)min(tm * BM2 + srow + 192, mlast) * g.lda + sseg * 2;
  // columns past the edge re-read the last pair (computed, never stored)
  const double* pB0 = g.B + (int64_t)krow * g.ldb + min(tn * BN + 2 * nseg, g.N - 2);
  const double* pB1 = pB0 + 8 * g.ldb;
  const int64_t bstep = (int64_t)BK * g.ldb;

  double2 xa0, xa1, xa2, xa3, xb0, xb1, ya0, ya1, ya2, ya3, yb0, yb1;
#define ISDF_NN_LOAD(C, A0, A1, A2, A3, B0, B1)                                               \
  {                                                                                           \
    const int off = (C) * BK;                                                                 \
    const int64_t offb = (C) * bstep;                                                         \
    A0 = *reinterpret_cast<const double2*>(pA0 + off);                                        \
    A1 = *reinterpret_cast<const double2*>(pA1 + off);                                        \
    A2 = *reinterpret_cast<const double2*>(pA2 + off);                                        \
    A3 = *reinterpret_cast<const double2*>(pA3 + off);                                        \
    B0 = *reinterpret_cast<const double2*>(pB0 + offb);                                       \
    B1 = *reinterpret_cast<const double2*>(pB1 + offb);                                       \
  }
#define ISDF_NN_ST1(P, R) { double* q_ = (P); q_[0] = R.x; q_[1] = R.y; }
#define ISDF_NN_STORE(BUF, A0, A1, A2, A3, B0, B1)                                            \
  {                                                                                           \
    double* qa = sA + (BUF) * BM2 * LDT + srow * LDT + sseg * 2;                              \
    double* qb = sB + (BUF) * BK * LDN + krow * LDN + 2 * nseg;                               \
    ISDF_NN_ST1(qa, A0) ISDF_NN_ST1(qa + 64 * LDT, A1) ISDF_NN_ST1(qa + 128 * LDT, A2)        \
    ISDF_NN_ST1(qa + 192 * LDT, A3)                                                           \
    *reinterpret_cast<double2*>(qb) = B0;                                                     \
    *reinterpret_cast<double2*>(qb + 8 * LDN) = B1;                                           \
  }

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  const int frow = lane & 15, fk = lane >> 4;
  const int aoff = (wm * 64 + frow) * LDT + fk, boff = fk * LDN + wn * 64 + frow;
  double a0[4], b0[4], a1[4], b1[4];
#define ISDF_NN_FRAG(BUF, KK, AF, BF)                                                         \
  {                                                                                           \
    const double* pa = sA + (BUF) * BM2 * LDT + aoff;                                         \
    const double* pb = sB + (BUF) * BK * LDN + boff;                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
      AF[i] = pa[i * 16 * LDT + (KK) * 4];                                                    \
      BF[i] = pb[(KK) * 4 * LDN + i * 16];                                                    \
    }                                                                                         \
  }
#define ISDF_NN_MFMA(AF, BF)                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(AF[i], BF[j], acc[i][j], 0, 0, 0);
  // same issue order as variant B (see there): LDS reads / writes / global loads trickle in after every second MFMA, one
  // barrier per chunk after k-step 2, the next chunk's first fragments prefetched under k-step 3
#define ISDF_NN_IL() _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#define ISDF_NN_IL_L() _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                \
    if (q_ < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#define ISDF_NN_IL_RW() _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                \
    if (q_ < 6) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#define ISDF_NN_CHUNK(BUF, LOAD_AHEAD, STORE_NEXT)                                            \
  {                                                                                           \
    ISDF_NN_FRAG(BUF, 1, a1, b1)                                                              \
    LOAD_AHEAD                                                                                \
    ISDF_NN_MFMA(a0, b0)                                                                      \
    ISDF_NN_IL_L()                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    ISDF_NN_FRAG(BUF, 2, a0, b0)                                                              \
    ISDF_NN_MFMA(a1, b1)                                                                      \
    ISDF_NN_IL()                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    ISDF_NN_FRAG(BUF, 3, a1, b1)                                                              \
    STORE_NEXT                                                                                \
    ISDF_NN_MFMA(a0, b0)                                                                      \
    ISDF_NN_IL_RW()                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    __syncthreads();                                                                          \
    ISDF_NN_FRAG(1 - (BUF), 0, a0, b0)                                                        \
    ISDF_NN_MFMA(a1, b1)                                                                      \
    ISDF_NN_IL()                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  }

  ISDF_NN_LOAD(0, xa0, xa1, xa2, xa3, xb0, xb1)
  ISDF_NN_LOAD(1, ya0, ya1, ya2, ya3, yb0, yb1)
  ISDF_NN_STORE(0, xa0, xa1, xa2, xa3, xb0, xb1)
  __syncthreads();
  ISDF_NN_FRAG(0, 0, a0, b0)
  __builtin_amdgcn_sched_barrier(0);
  for (int c = 0; c < nchunks; c += 2) {
    ISDF_NN_CHUNK(0, ISDF_NN_LOAD(min(c + 2, last), xa0, xa1, xa2, xa3, xb0, xb1), ISDF_NN_STORE(1, ya0, ya1, ya2, ya3, yb0, yb1))
    ISDF_NN_CHUNK(1, ISDF_NN_LOAD(min(c + 3, last), ya0, ya1, ya2, ya3, yb0, yb1), ISDF_NN_STORE(0, xa0, xa1, xa2, xa3, xb0, xb1))
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * BM2 + wm * 64 + i * 16 + (lane >> 4) + 4 * r;
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = tn * BN + wn * 64 + j * 16 + (lane & 15);
        if (col >= g.N) continue;
        const double v = acc[i][j][r];
        g.C[(int64_t)row * g.ldc + col] = SQ ? v * v : v;
      }
    }
  }
}

__global__ void reduce_slabs_kernel(const double* __restrict__ P, int nslab, int64_t slab_stride,
                                    int M, int N, double alpha, double beta, double* __restrict__ C,
                                    int64_t ldc) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx % N);
  double s = 0.0;
  for (int t = 0; t < nslab; ++t) s += P[(int64_t)t * slab_stride + idx];
  double* q = C + (int64_t)m * ldc + n;
  *q = (beta == 0.0) ? alpha * s : alpha * s + beta * (*q);
}

}  // namespace

int gemm_nt_f64_scaled(isdf_handle h, int M, int N, int64_t K, double alpha, const double* A, int64_t lda,
                       const double* B, int64_t ldb, const double* kscale, double beta, double* C,
                       int64_t ldc) {
  ARG_CHECK(h, M > 0 && N > 0 && K > 0 && A && B && C && lda >= K && ldb >= K && ldc >= N);
  GemmArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.kscale = kscale;
  g.M = M; g.N = N; g.K = K;
  // variant: -1 auto (default), 0 force the 128x128 kernel, 1 force the 256x128 kernel where legal, 3 the 128x128 kernel with
  // A direct to registers (variant D) where legal
  static const int variant = getenv("ISDF_GEMM_VARIANT") ? atoi(getenv("ISDF_GEMM_VARIANT")) : -1;
  const bool alignedB = (lda % 2 == 0) && (ldb % 2 == 0) && (K % 32 == 0) && (((uintptr_t)A) % 16 == 0) &&
                        (((uintptr_t)B) % 16 == 0) && (!kscale || ((uintptr_t)kscale) % 16 == 0);
  // auto: the 256x128 kernel (B) whenever M fills 256-row tiles (row padding < 15 %): after the issue-order work it leads on
  // every such shape measured, the k-point W^q batches most of all (MgO 2x2x2: 3.8 s against 4.5 s with D); variant D (A
  // direct to registers, 128x128, two workgroups per CU) for the other aligned shapes, variant A for unaligned operands
  const bool fitsB = (double)(cdiv(M, BM2) * BM2) <= 1.15 * (double)M;
  const bool useB = alignedB && M > BM && (variant == 1 || (variant == -1 && fitsB));
  const bool useD = alignedB && !useB && (variant == 3 || variant == -1);
  g.ntm = (int)cdiv(M, useB ? BM2 : BM); g.ntn = (int)cdiv(N, BN);
  const int64_t ntiles = (int64_t)g.ntm * g.ntn;
  // enough units to fill 2 workgroups per CU about 8 times over, slabs at least 2048 deep,
  // partial workspace at most 2 GiB
  const int64_t slots = (int64_t)h->num_cu * (useB ? 1 : 2);
  int64_t nslab = cdiv(8 * slots, ntiles);
  const int64_t cap = std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(1, K / 2048),
                                                             std::max<int64_t>(1, ((int64_t)2 << 30) / ((int64_t)M * N * 8))));
  nslab = std::max<int64_t>(1, std::min<int64_t>(nslab, cap));
  {
    // wave quantisation: the units are dealt to `slots` resident workgroups, so the launch takes ceil(units / slots) unit
    // times.  Among slab counts between the target and twice the target pick the one whose last round is fullest
    // (512 x 15132 output: 238 tiles; 9 slabs = 8.37 rounds -> 9, 15 slabs = 13.95 rounds -> 14: 7 % less idle time)
    static const int tune = getenv("ISDF_GEMM_SLAB_TUNE") ? atoi(getenv("ISDF_GEMM_SLAB_TUNE")) : 1;
    if (tune && ntiles * nslab > slots) {
      double best = 1e30;
      int64_t pick = nslab;
      for (int64_t c = nslab; c <= std::min<int64_t>(2 * nslab, cap); ++c) {
        const double units = (double)ntiles * c;
        const double waste = (double)(cdiv((int64_t)units, slots) * slots) / units;
        if (waste < best - 1e-9) { best = waste; pick = c; }
      }
      nslab = pick;
    }
  }
  g.kslab = cdiv(cdiv(K, nslab), 2 * BK) * (2 * BK);   // even number of chunks per slab
  g.nslab = (int)cdiv(K, g.kslab);
  g.nunits = ntiles * g.nslab;
  g.nunits_pad = cdiv(g.nunits, 8) * 8;
  g.alpha = alpha; g.beta = beta;
  if (g.nslab == 1) {
    g.direct = 1; g.P = C; g.ldp = ldc; g.slab_stride = 0;
  } else {
    g.direct = 0;
    g.P = (double*)isdf_ws(h, "gemm_partials", sizeof(double) * (size_t)g.nslab * M * N);
    if (!g.P) return ISDF_ERR_HIP;
    g.ldp = N; g.slab_stride = (int64_t)M * N;
  }
  const bool fast = (lda % 2 == 0) && (ldb % 2 == 0) && (K % BK == 0) &&
                    (((uintptr_t)A) % 16 == 0) && (((uintptr_t)B) % 16 == 0) &&
                    (!kscale || ((uintptr_t)kscale) % 16 == 0);
  ARG_CHECK(h, g.nunits_pad < 2147483647LL);
  // one profiling label per kernel instantiation, named as rocprofv3 names them
  const char* label = useD ? (kscale ? "gemm_nt_mfma_kernel_d<true>[flop]" : "gemm_nt_mfma_kernel_d<false>[flop]") : useB ? (kscale ? "gemm_nt_mfma_kernel_b<true>[flop]" : "gemm_nt_mfma_kernel_b<false>[flop]")
                           : (fast ? "gemm_nt_mfma_kernel<true>[flop]" : "gemm_nt_mfma_kernel<false>[flop]");
  {
    ProfScope ps(h, label, 2.0 * M * N * (double)K);
    if (useD) {
      if (kscale) hipLaunchKernelGGL(gemm_nt_mfma_kernel_d<true>, dim3((unsigned)g.nunits_pad), dim3(TPBD), 0, h->stream, g);
      else hipLaunchKernelGGL(gemm_nt_mfma_kernel_d<false>, dim3((unsigned)g.nunits_pad), dim3(TPBD), 0, h->stream, g);
    } else if (useB) {
      const size_t lds = sizeof(double) * 2 * (BM2 + BN) * LDT;
      if (!h->attr_gemm_b) {                       // per handle (= per device), not per process
        HIP_TRY(h, hipFuncSetAttribute((const void*)gemm_nt_mfma_kernel_b<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)gemm_nt_mfma_kernel_b<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        h->attr_gemm_b = 1;
      }
      if (kscale) hipLaunchKernelGGL(gemm_nt_mfma_kernel_b<true>, dim3((unsigned)g.nunits_pad), dim3(TPB2), lds, h->stream, g);
      else hipLaunchKernelGGL(gemm_nt_mfma_kernel_b<false>, dim3((unsigned)g.nunits_pad), dim3(TPB2), lds, h->stream, g);
    } else if (fast) hipLaunchKernelGGL(gemm_nt_mfma_kernel<true>, dim3((unsigned)g.nunits_pad), dim3(TPB), 0, h->stream, g);
    else hipLaunchKernelGGL(gemm_nt_mfma_kernel<false>, dim3((unsigned)g.nunits_pad), dim3(TPB), 0, h->stream, g);
    KERNEL_CHECK(h);
  }
  if (!g.direct) {
    ProfScope ps(h, "gemm_reduce_slabs_kernel[byte]", 8.0 * (double)M * N * (g.nslab + 1));
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)cdiv((int64_t)M * N, 256)), dim3(256), 0, h->stream,
                       g.P, g.nslab, g.slab_stride, M, N, alpha, beta, C, ldc);
    KERNEL_CHECK(h);
  }
  return ISDF_OK;
}

int gemm_nt_f64(isdf_handle h, int M, int N, int64_t K, double alpha, const double* A, int64_t lda,
                const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  return gemm_nt_f64_scaled(h, M, N, K, alpha, A, lda, B, ldb, nullptr, beta, C, ldc);
}

// Exposed for tests and benchmarks (declared in include/mi355_isdf.h).
extern "C" int isdf_gemm_nt(isdf_handle h, int M, int N, int64_t K, double alpha, const double* d_A,
                            int64_t lda, const double* d_B, int64_t ldb, const double* d_kscale,
                            double beta, double* d_C, int64_t ldc) {
  if (!h) return ISDF_ERR_ARG;
  return gemm_nt_f64_scaled(h, M, N, K, alpha, d_A, lda, d_B, ldb, d_kscale, beta, d_C, ldc);
}

bool gemm_nn_f64_supported(isdf_handle h, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B,
                           int64_t ldb) {
  // opt-in (isdf_set_option "gemm_nn_own"): measured on the pair-row shapes of configs[2] the kernel reaches 69.5-71 TF/s
  // alone and 66 TF/s inside the build, rocBLAS 74-75 and 74 (profiles/r02_gemm_nn_vs_rocblas.log) - the library keeps this
  // product by default
  return h->gemm_nn_own && M > BM && (double)(cdiv(M, BM2) * BM2) <= 1.15 * (double)M && N >= 2 && N % 2 == 0 && K >= 32 && K % 32 == 0 &&
         lda % 2 == 0 && ldb % 2 == 0 && ((uintptr_t)A) % 16 == 0 && ((uintptr_t)B) % 16 == 0 && M < 2147483647LL &&
         N < 2147483647LL && K < 2147483647LL;
}

int gemm_nn_f64(isdf_handle h, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B, int64_t ldb,
                double* C, int64_t ldc, bool square) {
  ARG_CHECK(h, A && B && C && lda >= K && ldb >= N && ldc >= N && gemm_nn_f64_supported(h, M, N, K, A, lda, B, ldb));
  GemmNNArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.ntm = (int)cdiv(M, BM2); g.ntn = (int)cdiv(N, BN);
  g.stm = g.ntm >= 4 ? 4 : (g.ntm >= 2 ? 2 : 1);
  g.stn = 32 / g.stm;
  g.ngm = (int)cdiv(g.ntm, g.stm);
  g.nunits = (int64_t)g.ngm * cdiv(g.ntn, g.stn) * 32;
  g.nunits_pad = cdiv(g.nunits, 8) * 8;
  ARG_CHECK(h, g.nunits_pad < 2147483647LL);
  const size_t lds = sizeof(double) * 2 * (BM2 * LDT + BK * LDN);
  if (!h->attr_gemm_nn) {
    HIP_TRY(h, hipFuncSetAttribute((const void*)gemm_nn_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(h, hipFuncSetAttribute((const void*)gemm_nn_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->attr_gemm_nn = 1;
  }
  ProfScope ps(h, square ? "gemm_nn_mfma_kernel<true>[flop]" : "gemm_nn_mfma_kernel<false>[flop]", 2.0 * M * N * (double)K);
  if (square) hipLaunchKernelGGL(gemm_nn_mfma_kernel<true>, dim3((unsigned)g.nunits_pad), dim3(TPB2), lds, h->stream, g);
  else hipLaunchKernelGGL(gemm_nn_mfma_kernel<false>, dim3((unsigned)g.nunits_pad), dim3(TPB2), lds, h->stream, g);
  KERNEL_CHECK(h);
  return ISDF_OK;
}

// Internal shared definitions for libmi355_isdf.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <hipfft/hipfft.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include "../../include/mi355_isdf.h"

struct FftPlan {
  hipfftHandle fwd = 0, bwd = 0;
  size_t work_bytes = 0;
};

// Optional in-library profiling: HIP event pairs on the work stream around named kernel launches,
// with the algorithmic work (bytes or flops) each launch stands for.  Off by default.
struct ProfRec {
  hipEvent_t e0 = nullptr, e1 = nullptr;
};
struct ProfEntry {
  std::vector<ProfRec> recs;
  double work = 0.0;        // summed algorithmic work
  double ms = 0.0;          // resolved on query
  int64_t launches = 0;
};

// exxdiv='vcut_ws' (pyscf/pbc/tools/pbc.py:318-346): parameters of the Wigner-Seitz truncated kernel; alpha = 0: off
struct WsKernel {
  double alpha = 0.0;
  double ak[9] = {0};        // lattice of the nk-fold cell (rows)
  int mesh[3] = {0, 0, 0};   // mesh of the precomputed table
  double maxq[3] = {0, 0, 0};
  const double* vq = nullptr;  // device table, mesh[0]*mesh[1]*mesh[2] doubles (caller-owned)
};

struct isdf_ctx {
  int profiling = 0;
  std::map<std::string, ProfEntry> prof;
  int device = 0;
  hipStream_t stream = nullptr;
  rocblas_handle blas = nullptr;
  std::string err;
  // cached FFT plans keyed by (n0,n1,n2,batch)
  std::map<std::vector<int>, FftPlan> plans;
  // named workspace buffers (grow-only until isdf_release_workspace)
  std::map<std::string, std::pair<void*, size_t>> ws;
  int num_cu = 256;
  // triangular solves of the fit: 0 = rocBLAS dtrsm (default, faster), 1 = substitution blocks of trsm.hip
  int trsm_substitution = 0;
  // Coulomb convolution: 2 = the hand-written FFT of fft_conv.hip in its three-pass plane form where the mesh allows, else its
  // five-pass form (default); 1 = always the five-pass form; 0 = hipFFT
  int own_fft = 2;
  // pair-density rows aoP ao: 0 = rocBLAS dgemm (default: 74 TF/s on that shape), 1 = the own MFMA NN kernel of gemm_f64.hip with
  // the square fused into its epilogue (66-71 TF/s; kept as the library-free route and for A/B runs)
  int gemm_nn_own = 0;
  int attr_gemm_b = 0, attr_gemm_nn = 0;   // dynamic-LDS attributes raised on this handle's device
  int conv_pipe = 1;         // plane passes of the convolution: 1 persistent workgroups with the next plane prefetched, 0 one workgroup per plane
  int conv_sub_rows = 0;     // rows per cache-resident sub-batch of the plane convolution (0: whole batch)
  int gram_pivot_tpb = 256;  // columns per workgroup of the Gram selection's pivot step (64, 128 or 256: measured 15.5 / 13.0 / 12.2 us per pivot)
  int block_apply_waves = 16;// grid of the register block apply: workgroups ~ this many times the CU count
  int block_apply_reg = 1;   // block apply with the block inverse in registers, persistent over column tiles (trsm.hip)
  // range-separation parameter of the Gamma-point Coulomb kernel table (0 = plain 1/r); isdf_set_coulomb_omega
  double coul_omega = 0.0;
  // spherical truncation radius of the Coulomb kernel (exxdiv='vcut_sph', pbc.py:312-317); 0 = none.  isdf_set_coulomb_cutoff
  double coul_rc = 0.0;
  int coul_sphere = 0;       // > 0: the Gamma-point kernel table keeps |G| <= coul_sphere percent of the inscribed sphere's radius
  WsKernel wsk;
};

int isdf_fail(isdf_handle h, int code, const char* fmt, ...);
void* isdf_ws(isdf_handle h, const char* name, size_t bytes);   // nullptr on failure (error set)
int isdf_get_plan(isdf_handle h, const int32_t mesh[3], int batch, FftPlan** out);
// coulomb.hip: the symmetrised half-spectrum Coulomb table of the handle's kernel state for this mesh, times extra_scale / prod(mesh)
// (workspace "coulG_half": valid until the next call)
int get_coulG_half(isdf_handle h, const int32_t mesh[3], const double a[9], double extra_scale, double** out);
// fft_conv.hip: d_out rows = ifft(cg * fft(d_in rows)) with the scaled half-spectrum table cg; zbuf nb * n0 n1 (n2/2+1) complex
bool conv_rows_own_supported(const int32_t mesh[3], int nb);
// k-point form (fft_conv.hip): d_re + i d_im rows = ifft(tab * fft(d_in rows)) with a full real table; zhalf nb * n0 n1 (n2/2+1),
// zfull nb * G complex scratch
bool conv_rows_q_own_supported(isdf_handle h, const int32_t mesh[3], int nb);
bool spectral_rows_own_supported(isdf_handle h, const int32_t mesh[3], int nb);
int spectral_rows_own(isdf_handle h, const double* d_in, int nb, const int32_t mesh[3], const int32_t* d_idx, const double* d_scale,
                      int npts, double* d_out, int64_t ldx, HIP_vector_type<double, 2u>* zbuf);
int conv_rows_q_own(isdf_handle h, const double* d_in, double* d_re, double* d_im, int nb, const int32_t mesh[3], const double* tab,
                    double2* zhalf, double2* zfull);
int conv_rows_own(isdf_handle h, const double* d_in, double* d_out, int nb, const int32_t mesh[3], const double* cg,
                  double2* zbuf);

#define HIP_TRY(h, expr)                                                                    \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return isdf_fail(h, ISDF_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,        \
                       hipGetErrorString(e_));                                              \
  } while (0)
#define BLAS_TRY(h, expr)                                                                   \
  do {                                                                                      \
    rocblas_status s_ = (expr);                                                             \
    if (s_ != rocblas_status_success)                                                       \
      return isdf_fail(h, ISDF_ERR_LIB, "%s:%d %s -> rocblas status %d", __FILE__, __LINE__, \
                       #expr, (int)s_);                                                     \
  } while (0)
#define FFT_TRY(h, expr)                                                                    \
  do {                                                                                      \
    hipfftResult r_ = (expr);                                                               \
    if (r_ != HIPFFT_SUCCESS)                                                               \
      return isdf_fail(h, ISDF_ERR_LIB, "%s:%d %s -> hipfft result %d", __FILE__, __LINE__,  \
                       #expr, (int)r_);                                                     \
  } while (0)
#define KERNEL_CHECK(h) HIP_TRY(h, hipGetLastError())
#define ARG_CHECK(h, cond)                                                                  \
  do {                                                                                      \
    if (!(cond)) return isdf_fail(h, ISDF_ERR_ARG, "%s:%d argument check failed: %s",       \
                                  __FILE__, __LINE__, #cond);                               \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// RAII marker: records event pairs around the launches issued during its lifetime when profiling is on.
struct ProfScope {
  isdf_handle h;
  ProfRec rec;
  ProfEntry* ent = nullptr;
  ProfScope(isdf_handle h_, const char* name, double work, int64_t launches = 1) : h(h_) {
    if (!h || !h->profiling) return;
    ent = &h->prof[name];
    ent->work += work;
    ent->launches += launches;
    if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess) { ent = nullptr; return; }
    (void)hipEventRecord(rec.e0, h->stream);
  }
  ~ProfScope() {
    if (!ent) return;
    (void)hipEventRecord(rec.e1, h->stream);
    ent->recs.push_back(rec);
  }
};

// ---- dense FP64 helpers implemented in gemm_f64.hip -------------------------------------------
// C (M x N row-major, ldc) = alpha * A (M x K row-major, lda) * B(N x K row-major, ldb)^T + beta * C
// "NT with K contiguous in both operands": the shape of W = V Theta^T and of J = ao (v.ao)^T.
int gemm_nt_f64(isdf_handle h, int M, int N, int64_t K, double alpha, const double* A, int64_t lda,
                const double* B, int64_t ldb, double beta, double* C, int64_t ldc);
// same with B scaled per k on the fly: C = alpha * A * (B .* kscale[None,:])^T + beta * C
int gemm_nt_f64_scaled(isdf_handle h, int M, int N, int64_t K, double alpha, const double* A, int64_t lda,
                       const double* B, int64_t ldb, const double* kscale, double beta, double* C,
                       int64_t ldc);
// Triangular solves by substitution (trsm.hip): L (m x m) lower triangular, row-major (reads the lower triangle only).
//   left:  X (m x n, row-major) <- op(L)^-1 X      right: X (n x m, row-major) <- X op(L)^-1      op = L | L^T (trans)
int trsm_lower_left(isdf_handle h, bool trans, int m, int64_t n, const double* L, int64_t ldl, double* X, int64_t ldx);
int trsm_lower_right(isdf_handle h, bool trans, int m, int64_t n, const double* L, int64_t ldl, double* X, int64_t ldx);
// every diagonal block of a block-diagonal lower factor in one launch: X[blk_b, :] <- D_b^-1 X[blk_b, :] (blk_off on the host)
int block_forward_solve(isdf_handle h, const double* D, int64_t ldd, int nblk, const int32_t* blk_off_host, double* X,
                        int64_t ldx, int64_t n);
// X[blk_b, :] <- Dinv_b X[blk_b, :] with explicit block inverses (lower triangular), MFMA kernel of trsm.hip
int block_apply_inverse(isdf_handle h, const double* Dinv, int64_t ldd, int nblk, const int32_t* blk_off_host, double* X,
                        int64_t ldx, int64_t n, bool square_input = false);
int transpose_rm(isdf_handle h, const double* src, int64_t lds, int64_t rows, int64_t cols, double* dst, int64_t ldd);
// The fit's triangular solves with a row-major lower factor L: dispatch on h->trsm_substitution between rocBLAS dtrsm
// (column-major view: the same buffer is the upper factor U = L^T) and trsm_lower_*.
int tri_left(isdf_handle h, bool trans, int m, int64_t n, const double* L, int64_t ldl, double* X, int64_t ldx);
int tri_right(isdf_handle h, bool trans, int m, int64_t n, const double* L, int64_t ldl, double* X, int64_t ldx);
// Row-major wrappers over rocBLAS for the well-shaped products.
// C (M x N, ldc) = alpha * op(A) * op(B) + beta * C, all row-major; opA/opB 'N' or 'T'.
// C = A B (or its element-wise square), A (M x K) and B (K x N) row-major, on the own MFMA NN kernel (gemm_f64.hip); callers test
// gemm_nn_f64_supported first and fall back to gemm_rm (+ a square pass)
bool gemm_nn_f64_supported(isdf_handle h, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B,
                           int64_t ldb);
int gemm_nn_f64(isdf_handle h, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B, int64_t ldb,
                double* C, int64_t ldc, bool square);
int gemm_rm(isdf_handle h, char opA, char opB, int64_t M, int64_t N, int64_t K, double alpha,
            const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
            int64_t ldc);

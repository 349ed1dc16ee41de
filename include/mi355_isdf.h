/*
 * mi355_isdf.h — C ABI of libmi355_isdf.so: the MI355X (gfx950) ISDF hot path for PySCF's
 * periodic density-fitting layer.
 *
 * Boundary style follows the reference's ctypes convention (SURVEY.md 8b "inner boundary"):
 * plain C, raw pointers + explicit sizes, caller-owned buffers, nothing returned by allocation
 * (pyscf/pbc/gto/eval_gto.py:140-151, pyscf/lib/numpy_helper.py:849-860, pyscf/pbc/tools/pbc.py:66-88).
 * Two deliberate differences: every entry point returns an int status (0 = ok; the message is
 * available from isdf_last_error), and data pointers named d_* are DEVICE pointers (HBM) so that
 * the stages chain without leaving the GPU.  Pointers without the d_ prefix are host pointers to
 * small tables.  All work is enqueued on the stream set with isdf_set_stream (default: the null
 * stream); calls are asynchronous with respect to the host unless stated otherwise.
 *
 * Layouts (double precision throughout, Γ point):
 *   ao      (nao, ld)   AO-major, grid index contiguous  — phi_mu(r_g) = ao[mu*ld + g]
 *                       (the transpose of the reference's (G, nao) eval_ao return, eval_gto.py:153-161)
 *   coords  (3, ngrids) SoA x|y|z                         (the reference passes F-ordered (G,3), eval_gto.py:130)
 *   theta   (P, ldt)    one interpolation vector per row, grid index contiguous (FFT batch layout)
 *   W       (P, ldw)    row-major
 *   dm, vj, vk (nset, nao, nao) row-major
 *   grid order: C order over (x,y,z) with fftfreq wrap-around, pyscf/pbc/gto/cell.py:874-898
 */
#ifndef MI355_ISDF_H
#define MI355_ISDF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct isdf_ctx* isdf_handle;

#define ISDF_OK 0
#define ISDF_ERR_ARG 1      /* bad argument / unsupported shape */
#define ISDF_ERR_HIP 2      /* HIP runtime error */
#define ISDF_ERR_LIB 3      /* rocBLAS / rocSOLVER / hipFFT error */
#define ISDF_ERR_NUM 4      /* numerical failure (e.g. Cholesky breakdown) */

/* ABI version of this header (bumped on any signature change). */
int isdf_abi_version(void);

/* Create/destroy a context bound to one GPU (one context per process/rank).
 * Replaces nothing in the reference (its native code is stateless); holds the hipFFT plans,
 * the rocBLAS/rocSOLVER handles and reusable workspace. */
int isdf_create(int device_id, isdf_handle* out);
int isdf_destroy(isdf_handle h);
int isdf_set_stream(isdf_handle h, void* hip_stream);
const char* isdf_last_error(isdf_handle h);
/* Bytes of device workspace currently held by the context. */
int64_t isdf_workspace_bytes(isdf_handle h);
/* Release cached workspace and FFT plans (keeps the context usable). */
int isdf_release_workspace(isdf_handle h);
/* Runtime switches.  "trsm_substitution": 0 (default) the fit's triangular solves with Cholesky factors go through rocBLAS
 * dtrsm, 1 through the substitution blocks of trsm.hip (plain forward/backward substitution on 64-row diagonal blocks +
 * dgemm updates: slower, no inverted diagonal blocks; an independent cross-check of rocBLAS's algorithm).  Unknown keys are
 * an error.  "own_fft": 2 (default) the Coulomb convolution runs through the hand-written FFT (fft_conv.hip): on 2-3-5 smooth
 * meshes whose (y, z) plane fits 160 KB of LDS in three passes (z and y transforms of a plane fused, x forward . kernel . x inverse
 * fused), otherwise - dimensions that factor into 2, 3, 5, 7, 11, 13 - in five streaming passes; 1 forces the five passes; 0
 * forces hipFFT (D2Z, kernel multiply, Z2D) everywhere (profiles/r02_conv_paths.log).
 * "gemm_nn_own": 0 (default) the pair-density rows phi_P^T phi go through rocBLAS dgemm (74 TF/s on that shape), 1 through the
 * hand-written MFMA NN kernel with the element-wise square in its epilogue (66-71 TF/s; profiles/r02_gemm_nn_vs_rocblas.log).
 * "block_apply_reg": 1 (default) isdf_block_apply keeps the block inverse in registers and walks the grid columns (blocks of up
 * to 256 rows), 0 the round-2 kernel that re-reads the inverse from L2 for every 32-column tile.
 * "conv_sub_rows": rows per sub-batch of the three-pass plane convolution (0, the default: the whole batch per pass); measured
 * neutral to slightly negative (profiles/r03_conv_sub_batches.log) - kept as the record of the experiment.
 * "conv_pipe": 1 (default) the plane passes of the three-pass convolution run as persistent workgroups with the next plane's
 * loads in flight under the current plane's stages (square planes of 64, 72, 80, 96, 100, 108 or 120 points a side), 0 one
 * workgroup per plane everywhere; same results bit for bit (profiles/r03_conv_pipelined_plane_passes.log).
 * "coul_sphere": p > 0 zeroes the Gamma-point kernel table beyond p percent of the radius of the sphere inscribed in the reciprocal
 * FFT box (0, the default: the whole box) - the experiment behind the spectral form of W (profiles/r03_sphere_check.log).
 * "gram_pivot_tpb": columns per workgroup of isdf_select_ip_gram's pivot step, 256 (default), 128 or 64; same pivots and
 * factor for every value, narrower is slower (12.2 / 13.0 / 15.5 us per pivot, profiles/r03_gram_pivot_step_widths.log). */
int isdf_set_option(isdf_handle h, const char* key, int value);
/* Range separation of the Gamma-point Coulomb kernel used by isdf_coulomb_W / _rows / _potential / isdf_get_j, as
 * pyscf/pbc/tools/pbc.py:408-418: omega > 0 long range (erf(omega r)/r), omega < 0 short range, 0 (default) plain 1/r.
 * (The k-point entry points take their kernel table from the caller.) */
int isdf_set_coulomb_omega(isdf_handle h, double omega);
/* Spherical truncation of the Coulomb kernel, exxdiv='vcut_sph' of the reference (pyscf/pbc/tools/pbc.py:312-317):
 * 4 pi / G^2 (1 - cos(|G| rc)), G = 0 -> 2 pi rc^2, with rc = (3 Nk vol / 4 pi)^(1/3) chosen by the caller; rc = 0 (default)
 * switches it off.  Applies to every kernel table built afterwards, isdf_coulG_q included; not combined with omega. */
int isdf_set_coulomb_cutoff(isdf_handle h, double rc);
/* Wigner-Seitz truncated kernel, exxdiv='vcut_ws' (pyscf/pbc/tools/pbc.py:318-346, table of :422-480): short-range part
 * 4 pi/|q+G|^2 (1 - exp(-|q+G|^2 / 4 alpha^2)) (zero vector: pi/alpha^2) + d_vq[index of q+G on the reciprocal lattice of
 * the nk-fold cell ak (rows), table mesh `mesh`] for |components of q+G| <= maxq.  d_vq is a caller-owned device table that
 * must outlive the setting; alpha <= 0 switches it off.  Applies to the tables built afterwards (isdf_coulG_q, the Gamma
 * half-spectrum table); takes precedence over omega / cutoff. */
int isdf_set_coulomb_ws(isdf_handle h, double alpha, const double ak[9], const int32_t mesh[3], const double maxq[3],
                        const double* d_vq);

/* Optional in-library profiling (bench.py's roofline leg): when enabled, the library brackets its
 * hot kernel launches with HIP events on the work stream and accumulates, per kernel name, the number
 * of launches, the elapsed milliseconds and the ALGORITHMIC work (bytes for HBM-bound kernels, flops
 * for MFMA-bound ones; DESIGN.md lists which).  isdf_prof_get synchronises the stream. */
int isdf_prof_enable(isdf_handle h, int on);
int isdf_prof_reset(isdf_handle h);
int isdf_prof_count(isdf_handle h);
int isdf_prof_get(isdf_handle h, int index, char* name, int name_cap, int64_t* launches,
                  double* total_ms, double* total_work);

/* S1. Periodic AO collocation, Γ point, real spherical GTOs, l <= 2.
 * Replaces PBCGTOval_sph_deriv0 (pyscf/lib/pbc/grid_ao.c:524-534, driver :439-486, per-shell image
 * loop :301-429; radial/angular parts pyscf/lib/gto/deriv1.c:31-165) as called from
 * pyscf/pbc/gto/eval_gto.py:140-151.  atm/bas/env are the libcint tables (pyscf/gto/mole.py:59-89),
 * Ls the translation list (eval_gto.py:132-136), rcut the per-shell cutoff (eval_gto.py:169-186).
 * Truncation rule: image T contributes to grid point r iff |r - R_atom - T| < rcut[shell]
 * (per point; the reference decides per block of 56 points — see oracle/ao.py). */
int isdf_eval_ao(isdf_handle h,
                 const int32_t* atm, int natm, const int32_t* bas, int nbas,
                 const double* env, int nenv,
                 const double* Ls, int nimgs, const double* rcut,
                 const double* d_coords, int64_t ngrids,
                 double* d_ao, int64_t ld);

/* Values and Cartesian first derivatives of the Gamma-point AOs (numint.eval_ao(deriv=1), pyscf/pbc/dft/numint.py:33-93;
 * pyscf/lib/gto/deriv1.c:60-69,166-330): four planes (value, d/dx, d/dy, d/dz) at d_ao + comp * plane_stride, each laid out
 * like isdf_eval_ao's output; same arguments and truncation rule otherwise.  For GGA densities and potentials. */
int isdf_eval_ao_deriv1(isdf_handle h, const int32_t* atm, int natm, const int32_t* bas, int nbas, const double* env,
                        int nenv, const double* Ls, int nimgs, const double* rcut, const double* d_coords,
                        int64_t ngrids, double* d_ao, int64_t ld, int64_t plane_stride);
/* k-point form: Bloch sums of values and Cartesian derivatives (times exp(-i k.r) with periodic_part - the derivative stays that
 * of the Bloch function), real and imaginary planes d_re / d_im + comp * plane_stride. */
int isdf_eval_ao_k_deriv1(isdf_handle h, const int32_t* atm, int natm, const int32_t* bas, int nbas, const double* env,
                          int nenv, const double* Ls, int nimgs, const double* rcut, const double kpt[3], int periodic_part,
                          const double* d_coords, int64_t ngrids, double* d_re, double* d_im, int64_t ld,
                          int64_t plane_stride);

/* k-point collocation: real and imaginary planes (nao rows each, leading dimension ld) of
 *   periodic_part = 0:  phi^k_m(r) = sum_T exp(i k.T) phi_m(r - T)       (eval_gto.py:137, grid_ao.c:421-422)
 *   periodic_part = 1:  u^k_m(r) = exp(-i k.r) phi^k_m(r)                 (lattice periodic)
 * Same tables, truncation rule and arithmetic as isdf_eval_ao. */
int isdf_eval_ao_k(isdf_handle h,
                   const int32_t* atm, int natm, const int32_t* bas, int nbas,
                   const double* env, int nenv,
                   const double* Ls, int nimgs, const double* rcut,
                   const double kpt[3], int periodic_part,
                   const double* d_coords, int64_t ngrids,
                   double* d_re, double* d_im, int64_t ld);

/* Copy columns: d_dst[mu*ld_dst + i] = d_src[mu*ld_src + d_idx[i]], i < n  (block-major regrouping
 * of the grid for local selection; also picks phi at interpolation points). */
/* d_out (nrow, nblk) = max |d_src[row, d_blk_off[b] .. d_blk_off[b+1])| (d_blk_off: nblk + 1 device int64 offsets): which AO rows
 * are identically zero on which block of grid points - the collocation truncates every shell at its rcut (eval_gto.py:169-186), so the
 * candidate selection of a block may skip those rows without changing a bit of its result. */
int isdf_block_row_absmax(isdf_handle h, const double* d_src, int nrow, int64_t ld, int nblk, const int64_t* d_blk_off,
                          double* d_out);
int isdf_gather_cols(isdf_handle h, const double* d_src, int nrow, int64_t ld_src,
                     const int64_t* d_idx, int64_t n, double* d_dst, int64_t ld_dst);

/* Voronoi partition of the grid for the per-atom selection blocks: d_owner[g] = index of the atom nearest to grid point g
 * (d_coords (3, ngrids) SoA) under the minimum-image convention over the 27 neighbouring images; among atoms within tie_atol
 * (Bohr) of the smallest distance the lowest index wins.  atom_coords (natm, 3) and the lattice a are host tables. */
int isdf_partition_by_atom(isdf_handle h, const double* d_coords, int64_t ngrids, const double* atom_coords, int natm,
                           const double a[9], double tie_atol, int32_t* d_owner);

/* S2. Interpolation-point selection: pivoted Cholesky of the implicit pair-density Gram matrix
 * A(r,r') = (sum_mu ao[mu,r] ao[mu,r'])^2, independently for nblk column blocks
 * [blk_off[b], blk_off[b+1]) of d_ao, nip[b] pivots each (stops early when the largest residual
 * diagonal <= tol; tol < 0 means m_b * eps * max diag).  Pivot rule: the reference's
 * pivoted_cholesky_python (pyscf/lib/scipy_helper.py:71-110) with a deterministic tie rule —
 * the lowest index whose residual >= (1 - tie_rtol) * max is taken (tie_rtol = 0: plain argmax).
 * Outputs: d_piv (nblk, kmax) int64 column indices LOCAL to each block, d_L (kmax, ldL) the
 * Cholesky rows (row j of block b in columns blk_off[b]..), rank[b] (host) pivots actually taken.
 * kmax = max_b nip[b].  Synchronises the stream before returning (rank is a host output). */
int isdf_select_ip(isdf_handle h, const double* d_ao, int nao, int64_t ld,
                   int nblk, const int64_t* blk_off, const int32_t* nip,
                   double tol, double tie_rtol,
                   double* d_L, int64_t ldL, int64_t* d_piv, int32_t* rank);

/* S2, refined stage: pivoted Cholesky of an EXPLICIT symmetric positive semidefinite matrix d_A (m x m, row-major,
 * leading dimension ldA, DESTROYED: it ends as the residual matrix) — the arithmetic of the reference's
 * pivoted_cholesky_python (pyscf/lib/scipy_helper.py:71-110) applied to the matrix itself, organised in panels of
 * ``panel`` pivots (<= 256; <= 0: 256) with a trailing update A <- A - Lp^T Lp after each panel (LAPACK dpstrf's scheme).
 * Used on the pair-density Gram matrix restricted to a candidate set of grid points (isdf_gram_sq of the candidates'
 * AO values): the candidates come from the per-atom selections, this call picks the final nip points among them.
 * Same stopping rule (tol < 0: m * eps * max diag) and tie rule as isdf_select_ip.  d_piv (nip) int64 receives the
 * pivots (indices into the candidate list), *rank (host) their number.  Synchronises the stream before returning. */
int isdf_select_ip_gram(isdf_handle h, double* d_A, int m, int64_t ldA, int nip, double tol, double tie_rtol,
                        int panel, int64_t* d_piv, int32_t* rank);

/* Complex (k-point) mode of S2: d_ao holds the lattice-periodic parts u^k_m of all Bloch AOs as
 * 2*nh real rows, rows [0,nh) = Re u, rows [nh,2nh) = Im u (nao = 2*nh, nh = nk*nao_cell).  The
 * Gram matrix is A(r,r') = |sum_m conj(u_m(r)) u_m(r')|^2 (real).  nh = 0 is isdf_select_ip. */
int isdf_select_ip_cplx(isdf_handle h, const double* d_ao, int nao, int nh, int64_t ld,
                        int nblk, const int64_t* blk_off, const int32_t* nip,
                        double tol, double tie_rtol,
                        double* d_L, int64_t ldL, int64_t* d_piv, int32_t* rank);

/* S3a. Fit from the selection's own factor (single block): Theta = T^-1 L in place, with
 * T[t,s] = L[t, piv[s]] upper triangular.  Equals the least-squares fit A_PP^-1 A_P. */
int isdf_fit_from_chol(isdf_handle h, double* d_L, int k, int64_t m, int64_t ldL,
                       const int64_t* d_piv);

/* S3b. Global least-squares fit for an arbitrary interpolation-point set d_ip (grid indices):
 * Theta = [(aoP aoP^T)^2 + reg*max(diag)*I]^-1 (aoP ao)^2 by Cholesky.
 * reg_rel >= 0 is the requested relative diagonal shift; if the factorisation breaks down the shift
 * is raised (1e-14, then x100 per retry, 4 retries) and the value finally used is returned in
 * *reg_used (host; may be NULL).  ISDF_ERR_NUM if it still fails.
 *   isdf_fit_prepare: d_aoP (P, nao) = phi at the points, d_chol (P, P) = Cholesky factor.
 *                     Synchronises the stream (breakdown is reported to the host).
 *   isdf_fit_apply:   Theta columns for any slice of the grid: d_ao points at the slice's first
 *                     column (nao, ld), ng columns; d_theta likewise (P, ldt).  Grid slices are
 *                     independent, which is what grid-sharded multi-GPU runs use.
 *                     forward_only != 0 stops after the forward solve and returns Y = Lr^-1 B
 *                     (A_PP = Lr Lr^T) instead of Theta = Lr^-T Y: half the flops.  W is then
 *                     obtained as Lr^-T [w conv(Y) Y^T] Lr^-1 by isdf_W_from_factor — the same W in
 *                     exact arithmetic (DESIGN.md section 2 discusses the conditioning).
 *   isdf_fit_global:  both, on the whole grid, factor kept in the context workspace. */
int isdf_fit_prepare(isdf_handle h, const double* d_ao, int nao, int64_t ld,
                     const int64_t* d_ip, int P, double reg_rel, double* d_aoP, double* d_chol,
                     double* reg_used);
int isdf_fit_apply(isdf_handle h, const double* d_chol, const double* d_aoP, int P, int nao,
                   const double* d_ao, int64_t ng, int64_t ld, int forward_only,
                   double* d_theta, int64_t ldt);
/* Complex (k-point) mode of the fit, d_ao = [Re u; Im u] as in isdf_select_ip_cplx:
 * A_PP and B are |S|^2 = (aoP X)^2 + (aoP_rot X)^2 with aoP_rot = [Im u_P | -Re u_P]; Theta is real. */
int isdf_fit_prepare_cplx(isdf_handle h, const double* d_ao, int nao, int nh, int64_t ld,
                          const int64_t* d_ip, int P, double reg_rel, double* d_aoP, double* d_chol,
                          double* reg_used);
int isdf_fit_apply_cplx(isdf_handle h, const double* d_chol, const double* d_aoP, int P, int nao, int nh,
                        const double* d_ao, int64_t ng, int64_t ld, int forward_only,
                        double* d_theta, int64_t ldt);
int isdf_fit_global(isdf_handle h, const double* d_ao, int nao, int64_t ngrids, int64_t ld,
                    const int64_t* d_ip, int P, double reg_rel, double* d_theta, int64_t ldt,
                    double* d_aoP, double* reg_used);

/* S3c. Block-Jacobi route — no triangular solve over the grid.  With A = A_PP, B = A_P and
 * D = blockdiag(chol(A_bb)) over the per-atom blocks of points [blk_off[b], blk_off[b+1]):
 *   Y' = D^-1 B,  M' = w conv(Y') Y'^T (isdf_coulomb_W),  A' = D^-1 A D^-T   (A already carries the fit's diagonal shift,
 *   isdf_shift_diag, so that S3b and S3c solve the same regularised normal equations),
 *   W  = D^-T [A'^-1 M' A'^-1] D^-1     ( = A^-1 [w conv(B) B^T] A^-1, the same W as S3a/S3b ).
 * Building blocks (nh > 0 selects the complex k-point mode of the Gram products):
 *   isdf_gather_aoP      d_aoP (P, nao) = ao[:, ip]^T
 *   isdf_gram_sq         d_A (P, P) = (aoP aoP^T)^2
 *   isdf_pair_gram_rows  d_B (P, ldb) = (aoP ao)^2 on ng grid columns
 *   isdf_block_chol      d_D (P, P): zero except the diagonal blocks, which hold the row-major lower Cholesky
 *                        factors D_b of A_bb + shift*max(diag A)*I; shift = shift_rel, raised per block (1e-14, x100 per
 *                        retry) when the block is not numerically positive definite - D is only a preconditioner;
 *                        *shift_used = the largest shift any block needed
 *   isdf_block_solve     side 0: X (P, n) <- op(D)^-1 X;  side 1: X (n, P) <- X op(D)^-1;  op = D (trans 0) | D^T (trans 1)
 *   isdf_chol_inplace    d_A <- Cholesky factor (same storage convention and the same shift ladder as
 *                        isdf_fit_prepare's d_chol; *reg_used = the relative shift that succeeded; d_scratch:
 *                        P*P doubles for the retries, NULL = library workspace)
 *   isdf_W_from_factor   kind 2: M <- U^-T M U^-1 ; followed by kind 0 gives A^-1 M A^-1. */
int isdf_gather_aoP(isdf_handle h, const double* d_ao, int nao, int64_t ld, const int64_t* d_ip, int P,
                    double* d_aoP);
int isdf_gram_sq(isdf_handle h, const double* d_aoP, int P, int nao, int nh, double* d_A);
int isdf_pair_gram_rows(isdf_handle h, const double* d_aoP, int P, int nao, int nh, const double* d_ao,
                        int64_t ng, int64_t ld, double* d_B, int64_t ldb);
int isdf_block_chol(isdf_handle h, const double* d_A, int P, int nblk, const int32_t* blk_off,
                    double shift_rel, double* d_D, double* shift_used);
int isdf_block_solve(isdf_handle h, const double* d_D, int P, int nblk, const int32_t* blk_off, int side,
                     int trans, double* d_X, int64_t n, int64_t ldx);
/* The block solves on the matrix cores: isdf_block_invert forms Dinv = blockdiag(D_b^-1) (P x P, lower triangular blocks, exact
 * zeros elsewhere) from the factors of isdf_block_chol; isdf_block_apply overwrites the rows of every block with
 * Dinv_b X_b (d_X: rows blk_off[0]..blk_off[nblk], n columns, leading dimension ldx; d_Dinv may point at a diagonal
 * sub-block, ldd its leading dimension) - the MFMA form of isdf_block_solve(side 0, trans 0): one read and one write of X. */
int isdf_block_invert(isdf_handle h, const double* d_D, int P, int nblk, const int32_t* blk_off, double* d_Dinv);
int isdf_block_apply(isdf_handle h, const double* d_Dinv, int64_t ldd, int nblk, const int32_t* blk_off,
                     double* d_X, int64_t n, int64_t ldx);
/* isdf_pair_gram_rows followed by isdf_block_apply with the square folded into the second: d_B (P, ng) <- Dinv_b (aoP ao)^2
 * for the P = blk_off[nblk] points of the given blocks (real mode). */
int isdf_pair_rows_block_apply(isdf_handle h, const double* d_aoP, int P, int nao, const double* d_ao, int64_t ng,
                               int64_t ld, const double* d_Dinv, int64_t ldd, int nblk, const int32_t* blk_off,
                               double* d_B, int64_t ldb);

/* The (AO x occupied orbital) pair space: what the reference's K works on when the density matrix carries its orbitals
 * (mo_coeff / mo_occ tag, pyscf/pbc/df/fft_jk.py:206-210,235-238: pair densities phi_mu psi_i, N x N_occ of them instead of
 * N(N+1)/2).  With psi (nocc, ng) = C_occ^T phi on the grid and psiP (P, nocc) its values at the points, the Gram matrix of
 * the pair products is the element-wise PRODUCT of two Gram matrices where isdf_gram_sq / isdf_pair_gram_rows have a square:
 *   isdf_gram_prod       d_A (P, P)   = (aoP aoP^T) o (psiP psiP^T)
 *   isdf_pair_prod_rows  d_B (P, ldb) = (aoP ao) o (psiP psi) on ng grid columns (ld / ldpsi: leading dimensions of ao / psi)
 *   isdf_factor_solve_half  X (P, n) <- L^-1 X (backward 0) | L^-T X (backward 1) for A = L L^T as stored by
 *                        isdf_chol_inplace: the Cholesky fit route on rows the caller produced with isdf_pair_prod_rows.
 * Selection (isdf_select_ip_gram on isdf_gram_prod of the candidates), block factors, W and K are the same calls as in the
 * AO x AO pair space. */
int isdf_gram_prod(isdf_handle h, const double* d_aoP, int P, int nao, const double* d_psiP, int nocc, double* d_A);
int isdf_pair_prod_rows(isdf_handle h, const double* d_aoP, int P, int nao, const double* d_psiP, int nocc,
                        const double* d_ao, int64_t ld, const double* d_psi, int64_t ldpsi, int64_t ng,
                        double* d_B, int64_t ldb);
int isdf_factor_solve_half(isdf_handle h, const double* d_fac, int P, int backward, double* d_X, int64_t n, int64_t ldx);

/* d_A <- d_A + shift_rel * max(diag d_A) * I. */
int isdf_shift_diag(isdf_handle h, double* d_A, int P, double shift_rel);
int isdf_chol_inplace(isdf_handle h, double* d_A, int P, double shift_rel, double* d_scratch, double* reg_used);

/* d_X (P, n) row-major, ldx >= n  <-  A^-1 X, A = U^T U the factor of isdf_fit_prepare / isdf_chol_inplace.  Column blocks are
 * independent: the grid-sharded build gives every rank n = P/R columns of the P x P finishing solves. */
int isdf_factor_solve(isdf_handle h, const double* d_fac, int P, double* d_X, int64_t n, int64_t ldx);

/* The two halves of isdf_bj_probe_rows, for builds that hold only a panel of the fit rows at a time:
 * isdf_bj_probe_vectors: d_T (n, P) rows t_j -> e_j = A'^-1 D^-1 t_j in place;
 * isdf_rows_combine: d_F (n, ng) (+)= d_E (n, rows; leading dimension ldE) d_Y (rows, ng) - for n <= 8 one streaming pass
 * over Y (HBM-bound), accumulate != 0 adds to F. */
int isdf_bj_probe_vectors(isdf_handle h, double* d_T, int n, const double* d_fac, const double* d_D, int P,
                          int nblk, const int32_t* blk_off);
int isdf_rows_combine(isdf_handle h, const double* d_E, int n, int64_t ldE, int rows, const double* d_Y,
                      int64_t ng, int64_t ldy, double* d_F, int64_t ldf, int accumulate);

/* A-posteriori check of the block-Jacobi route (it amplifies rounding in M' by cond(A')):
 * for probe vectors t_j (rows of d_T, values of a density at the points) the fitted density Theta^T t_j on ng grid
 * columns,   d_T (n, P) <- e_j = A'^-1 D^-1 t_j  (in place),   d_F (n, ldf) <- E Y'.
 * The caller compares  t^T W t  with  w * sum_g F conv(F)  (isdf_coulomb_rows + isdf_gemm_nt). */
int isdf_bj_probe_rows(isdf_handle h, double* d_T, int n, const double* d_fac, const double* d_D, int P, int nblk,
                       const int32_t* blk_off, const double* d_Yp, int64_t ng, int64_t ldy, double* d_F, int64_t ldf);

/* T (k, k) row-major upper triangular, T[t][s] = L[t][piv[s]] for s >= t: the triangular factor that
 * turns the selection's Cholesky rows into interpolation vectors (Theta = T^-1 L). */
int isdf_gather_T(isdf_handle h, const double* d_L, int k, int64_t ldL, const int64_t* d_piv,
                  double* d_T);
/* W = S^-1 M S^-T in place on d_M (P, ldm), where Theta = S^-1 Y and M = w conv(Y) Y^T:
 *   kind 0: d_F = Cholesky factor written by isdf_fit_prepare (S = Lr^T),  Y from isdf_fit_apply(forward_only)
 *   kind 1: d_F = T from isdf_gather_T (S = T),                           Y = L, the selection's rows.
 *   kind 2: d_F as kind 0, M <- U^-T M U^-1 (first half of A^-1 M A^-1, see S3c).
 * Replaces the second O(P^2 G) triangular solve by two O(P^3) ones. */
int isdf_W_from_factor(isdf_handle h, const double* d_F, int P, int kind, double* d_M, int64_t ldm);

/* S4+S5. Coulomb convolution and W:  for rows p in [row0, row0+nrows):
 *   V_p = ifft( coulG * fft(theta_p) ).real,   W[p, q] = (vol/G) * sum_g V_p[g] theta_q[g],  q < P.
 * FFT conventions of pyscf/pbc/tools/pbc.py:149-211 (forward unscaled, inverse 1/G), Coulomb kernel
 * 4 pi/|G|^2 with G=0 -> 0 (pbc.py:352-356) on the fftfreq-ordered mesh (cell.py:552-587),
 * normalisation of pyscf/pbc/df/fft_ao2mo.py:154-184.  a = lattice vectors (3,3 row-major, Bohr).
 * d_W rows [row0, row0+nrows) are written (row-major, ldw >= P).  batch = rows per FFT batch.
 * upper_only != 0: W is symmetric, so for each batch only the columns from the batch's first row on
 * are computed (half the flops); call isdf_symmetrize_upper once all rows are done. */
int isdf_coulomb_W(isdf_handle h, const double* d_theta, int P, int64_t ldt,
                   const int32_t mesh[3], const double a[9],
                   int row0, int nrows, int batch, int upper_only, double* d_W, int64_t ldw);
/* Spectral form of W.  By Parseval  w sum_r Theta_P(r) conv(Theta_Q)(r) = (w / G) sum_G coulG(G)
 * fft(Theta_P)(G) conj(fft(Theta_Q)(G)); the sum over the half spectrum with multiplicities (1 on the kz = 0 and Nyquist planes,
 * 2 between) is X X^T for the packed real rows X[r][2j], X[r][2j+1] = scale_j (Re, Im) fft(rows[r])[idx_j].  Restricted to the
 * points inside a sphere of the reciprocal FFT box the sum loses what the kernel table's corners carry (1e-9 Eh in E_K at BASELINE
 * configs[2], profiles/r03_sphere_check.log) and about half of its terms.
 *   isdf_coulG_half:         d_out (n0 n1 (n2/2+1)) = the symmetrised half-spectrum table the Gamma-point convolution multiplies with
 *                            (1/G inside; honours omega / cutoff / Wigner-Seitz state and the option "coul_sphere")
 *   isdf_spectral_supported: *ok = 1 when isdf_spectral_rows covers this mesh (2-3-5 smooth, (y, z) plane fits LDS)
 *   isdf_spectral_rows:      d_out (nrows, ldx; ldx even, >= 2 npts; padding zeroed) from nrows real rows (ld == G): forward
 *                            plane pass + forward x pass of the own FFT, then the gather; batch rows per pass */
int isdf_coulG_half(isdf_handle h, const int32_t mesh[3], const double a[9], double* d_out);
int isdf_spectral_supported(isdf_handle h, const int32_t mesh[3], int batch, int* ok);
int isdf_spectral_rows(isdf_handle h, const double* d_in, int nrows, int64_t ld, const int32_t mesh[3], const int32_t* d_idx,
                       const double* d_scale, int npts, int batch, double* d_out, int64_t ldx);
/* S4 alone: d_out rows = ifft(coulG * fft(d_in rows)).real (no weight), rows of length G contiguous
 * (ld == ldo == G); in place allowed.  Used by the grid-sharded multi-GPU path, where the rows are
 * assembled by an all-to-all before and scattered by another one after this call. */
int isdf_coulomb_rows(isdf_handle h, const double* d_in, int nrows, int64_t ld,
                      const int32_t mesh[3], const double a[9], int batch, double* d_out, int64_t ldo);
/* W[q][p] = W[p][q] for q > p. */
int isdf_symmetrize_upper(isdf_handle h, double* d_W, int P, int64_t ldw);
/* W <- (W + W^T) / 2, or (W - W^T) / 2 with antisymmetric != 0 (the imaginary plane of a Hermitian W^q) (after the two-sided solves of the block-Jacobi route, whose rounding along null(A_PP) is not
 * symmetric; the mean keeps that noise inside null(A_PP) x null(A_PP), mirroring one triangle would not). */
int isdf_symmetrize_mean(isdf_handle h, double* d_W, int P, int64_t ldw, int antisymmetric);

/* S6. J exactly as pyscf/pbc/df/fft_jk.py:63-107 (Γ, real dm):
 *   rho = sum_mn dm_mn ao_m ao_n;  v = (vol/G) ifft(coulG fft rho).real;  vj = ao (v .* ao)^T.
 * Grid columns [g0, g0+ng) only are contracted when ng < ngrids (grid-sharded partial sums:
 * the density/potential FFT still uses the full grid, so ng<ngrids requires d_vR_in/out staging —
 * see isdf_rho / isdf_vj_from_vR). */
int isdf_get_j(isdf_handle h, const double* d_ao, int nao, int64_t ngrids, int64_t ld,
               const int32_t mesh[3], const double a[9],
               const double* d_dm, int nset, double* d_vj);
/* The three pieces of S6 separately (for grid-sharded multi-GPU runs). */
int isdf_rho(isdf_handle h, const double* d_ao, int nao, int64_t ng, int64_t ld,
             const double* d_dm, int nset, double* d_rho, int64_t ldrho);
int isdf_coulomb_potential(isdf_handle h, double* d_rho_inout, int nset, int64_t ldrho,
                           const int32_t mesh[3], const double a[9]);
int isdf_vj_from_vR(isdf_handle h, const double* d_ao, int nao, int64_t ng, int64_t ld,
                    const double* d_vR, int nset, int64_t ldv, double* d_vj);

/* The reference's EXACT exchange on the device (pyscf/pbc/df/fft_jk.py:177-302, Gamma point, occupied-
 * orbital form :235-238,256-259): for AO rows i in [i0, i0+ni)
 *   vk[i, l] = (vol/G) sum_g [ sum_j ifft(coulG fft(ao_i mo_j))(g) mo_j(g) ] ao_l(g),   mo = C^T ao,
 * d_C (nao, nocc) = occupied MO coefficients times sqrt(occupation).  N*nocc FFT pairs — the cost
 * ISDF removes; provided to MEASURE the fitting error of the ISDF K at full size, not as the fast path.
 * max_rows = pair-density rows convolved per pass (memory: 24 * max_rows * G bytes). */
int isdf_get_k_exact(isdf_handle h, const double* d_ao, int nao, int64_t ngrids, int64_t ld,
                     const double* d_C, int nocc, const int32_t mesh[3], const double a[9],
                     int i0, int ni, int max_rows, double* d_vk);

/* S7. K from the interpolation factorisation:
 *   vk = aoP^T [ (aoP dm aoP^T) .* W ] aoP   for rows [row0,row0+nrows) of the Hadamard matrix
 * (row-sharded partial sums for multi-GPU; pass row0=0,nrows=P for the whole thing). */
int isdf_get_k(isdf_handle h, const double* d_aoP, int P, int nao,
               const double* d_W, int64_t ldw, int row0, int nrows,
               const double* d_dm, int nset, double* d_vk);

/* Coulomb kernel table for a difference vector q (Cartesian, 1/Bohr) on the full FFT mesh, fftfreq C order, G doubles:
 * 4 pi / |q + G|^2 with the reference's treatment of components beyond / on the mesh edge for q != 0 (wrap-around,
 * pyscf/pbc/tools/pbc.py:272-302, zeroed edge entries :400-401; wrap_around = 0 switches both off), |q + G| = 0 -> 0
 * (:352-356), range separation omega as :408-418 (0: plain).  Replaces tools.get_coulG(cell, k, exx=None) of the reference
 * as fft_jk.py:276 calls it; the table feeds isdf_coulomb_Wq.  Fails with ISDF_ERR_ARG when q lies outside the first FFT
 * box (the reference's assertion, pbc.py:281). */
int isdf_coulG_q(isdf_handle h, const int32_t mesh[3], const double a[9], const double q[3], int wrap_around,
                 double omega, double* d_out);

/* ---- k-points (BASELINE configs[3]); conventions of pyscf/pbc/df/fft_jk.py:177-302 ------------------
 * Bloch AOs are carried as lattice-periodic parts in two real planes (isdf_eval_ao_k, periodic_part=1);
 * Theta / Y rows are real and k-independent (complex modes of S2/S3).  Per difference vector
 * q = k2 - k1 (k1 a band k-point):
 *   isdf_coulomb_Wq:  rows [row0,row0+nrows) of  M^q = weight * V^q Theta^T,  V^q_P = ifft(coulG_q fft(Theta_P)),
 *                     d_coulG = the G real values of get_coulG(cell, q) (pbc/tools/pbc.py:230-420) in
 *                     fftfreq order; complex transform (Z2Z) because coulG(q+G) is not inversion
 *                     symmetric.  Real and imaginary parts in d_Wre / d_Wim; upper_only as in isdf_coulomb_W.
 *   isdf_symmetrize_hermitian:  M[q][p] = conj(M[p][q]) for q > p.
 *   (isdf_W_from_factor is applied to both planes.)
 *   isdf_finish_Wq:   d_Wc (P, P) complex interleaved = (Wre + i Wim)[p][q] * ph[p] * conj(ph[q]),
 *                     d_phase = (P, 2) with ph[p] = exp(-i q.r_p).
 *   isdf_get_k_pair:  d_vk (nao, nao) complex += scale * A1^H [ (A2 D2 A2^H) .* Wq ] A1 with
 *                     A1, A2 = phi^{k1}, phi^{k2} at the interpolation points, (P, nao) complex interleaved,
 *                     D2 (nao, nao) complex; rocBLAS zgemm + a fused complex Hadamard kernel.
 *   isdf_rho_k:       d_rho[g] += scale * sum_ij D_ij u_i(g) conj(u_j(g)) (real part; Hermitian D), with
 *                     d_DTr/d_DTi = real/imag planes of D^T (fft_jk.py:84-88).
 *   isdf_vj_k:        vj_ij = sum_g conj(u_i(g)) vR[g] u_j(g)  (fft_jk.py:100-107), planes re/im. */
int isdf_coulomb_Wq(isdf_handle h, const double* d_theta, int P, int64_t ldt, const int32_t mesh[3],
                    const double* d_coulG, double weight, int row0, int nrows, int batch,
                    int upper_only, double* d_Wre, double* d_Wim, int64_t ldw);
int isdf_symmetrize_hermitian(isdf_handle h, double* d_Wre, double* d_Wim, int P, int64_t ldw);
int isdf_finish_Wq(isdf_handle h, const double* d_Wre, const double* d_Wim, int P, int64_t ldw,
                   const double* d_phase, double* d_Wc);
int isdf_get_k_pair(isdf_handle h, const double* d_A1, const double* d_A2, const double* d_D2,
                    const double* d_Wq, int P, int nao, double scale, double* d_vk);

/* The reference's exact k-point exchange for one (k1, k2) pair on the device (pyscf/pbc/df/fft_jk.py:250-292: pair densities
 * conj(phi^{k1}_p) exp(-i q.r) phi^{k2}_j, kernel get_coulG(cell, k2 - k1), weight 1/nk vol/G) - the verification path of the
 * k-point ISDF exchange, N * nocc complex FFT pairs per (k1, k2).  In periodic parts the phases cancel:
 *   vk[p - i0, k'] += weight sum_g { sum_j conv_q[conj(u1_p) m2_j](g) conj(m2_j(g)) } u1_k'(g),   p in [i0, i0 + ni)
 * d_u1r/d_u1i (nao, ld1): periodic parts of the Bloch AOs at k1 (isdf_eval_ao_k, periodic_part = 1); d_m2r/d_m2i (nocc, ld2):
 * those of the occupied orbitals at k2 scaled with sqrt(occ); d_coulG (G): isdf_coulG_q of q = k2 - k1; max_rows: FFT rows per
 * pass (>= nocc); d_vk_re/d_vk_im (ni, nao) row-major are ACCUMULATED into (the caller sums over k2). */
int isdf_get_k_exact_kpt(isdf_handle h, const double* d_u1r, const double* d_u1i, int nao, int64_t ld1,
                         const double* d_m2r, const double* d_m2i, int nocc, int64_t ld2, const int32_t mesh[3],
                         const double* d_coulG, double weight, int i0, int ni, int max_rows, double* d_vk_re,
                         double* d_vk_im);

/* Robust K at k-points (Dunlap's correction with V^q = conv_q(Theta)): building blocks next to isdf_gemm_nn /
 * isdf_gemm_nt on the real / imaginary planes.
 *   isdf_coulomb_rows_q:   (d_re + i d_im)[r] = ifft(coulG(q) fft(rows[r])) for nrows real rows of G = prod(mesh) points (ld == G);
 *                          d_coulG from isdf_coulG_q; the convolution step of isdf_coulomb_Wq on its own
 *   isdf_zhadamard_planes: (Ar + i Ai) .*= (Br + i Bi) on `rows` x `cols` planes */
int isdf_coulomb_rows_q(isdf_handle h, const double* d_rows, int nrows, int64_t ld, const int32_t mesh[3],
                        const double* d_coulG, double* d_re, double* d_im);
/* (d_re + i d_im)[r] (mesh[a] x mesh[b] entries, the two axes other than `axis`, C order) = the 3-D DFT of the real row r on the
 * Nyquist plane of `axis` (mesh[axis] even): what the even-mesh correction of the +-q pairing needs (W^{-q} = conj(W^q) holds
 * index by index only off the Nyquist planes: pbc.py:272-302 labels index n/2 as -n/2 for both signs of q). */
int isdf_nyquist_spectra(isdf_handle h, const double* d_rows, int nrows, int64_t ld, const int32_t mesh[3], int axis,
                         double* d_re, double* d_im);
int isdf_zhadamard_planes(isdf_handle h, double* d_Ar, double* d_Ai, int64_t lda, const double* d_Br, const double* d_Bi,
                          int64_t ldb, int rows, int64_t cols);
int isdf_rho_k(isdf_handle h, const double* d_ur, const double* d_ui, int nao, int64_t ng, int64_t ld,
               const double* d_DTr, const double* d_DTi, double scale, double* d_rho);
int isdf_vj_k(isdf_handle h, const double* d_ur, const double* d_ui, int nao, int64_t ng, int64_t ld,
              const double* d_vR, double* d_vj_re, double* d_vj_im);

/* ---- GTH pseudopotential pieces of FFTDF.get_pp (pyscf/pbc/df/fft.py:64-152) -----------
 * isdf_pp_local_potential:  d_vlocR (G) = ifft( -sum_a exp(-i G.R_a) vloc_a(G) ).real with the GTH local
 *     form of pyscf/pbc/gto/pseudo/pp.py:58-93 + pp_int.py:51-71; atoms without a pseudopotential enter as
 *     bare nuclei.  coords (natm,3) Bohr; pp_par (natm,8) = [has_pp, Z, rloc, nexp, C1, C2, C3, C4] (host).
 *     The matrix elements follow with isdf_vj_from_vR / isdf_vj_k.
 * isdf_pp_projector_overlaps:  d_out (nrows, nao) complex = sum_G conj(SI_a(G)) p_j(G+k) aoG_mu(G+k), the
 *     non-local projector/AO overlaps of fft.py:88-131 with the analytic AO Fourier transform; proj_tab
 *     (nproj,3) = [atom, l, i], proj_rl (nproj) = r_l; rows are (projector j, m = -l..l) in order.
 *     The caller finishes with vnl = 1/vol sum conj(SPG_i) h_ij SPG_j (fft.py:132-140). */
int isdf_pp_local_potential(isdf_handle h, int natm, const double* coords, const double* pp_par,
                            const int32_t mesh[3], const double a[9], double* d_vlocR);
int isdf_pp_projector_overlaps(isdf_handle h, const int32_t* atm, int natm, const int32_t* bas, int nbas,
                               const double* env, int nenv, const double* coords, const double kpt[3],
                               const int32_t* proj_tab, const double* proj_rl, int nproj,
                               const int32_t mesh[3], const double a[9], double* d_out);

/* Robust-fitting K (Dunlap's correction on top of the ISDF exchange).  With V_P = conv(Theta_P) kept
 * on the device, per batch of points:  F (nb, G) = (phi_P D)(nb, N) . phi (N, G)  [isdf_gemm_nn],  F .*= V rows
 * [isdf_hadamard_rows],  (F phi^T)(nb, N) [isdf_gemm_nt]  ->  K1 = w phi_P^T (F phi^T);  K = K1 + K1^T - K_isdf.
 *   isdf_gemm_nn:       C (M, ldc) = alpha A (M, K; lda) B (K, N; ldb) + beta C, row-major, N contiguous (rocBLAS dgemm; with
 *                       option "gemm_nn_own" the own MFMA NN kernel for alpha = 1, beta = 0 on aligned operands with
 *                       K % 32 == 0 and full 256-row tiles)
 *   isdf_hadamard_rows: X (rows, ldx) .*= Y (rows, ldy) on `cols` columns */
int isdf_gemm_nn(isdf_handle h, int M, int64_t N, int K, double alpha, const double* d_A, int64_t lda, const double* d_B,
                 int64_t ldb, double beta, double* d_C, int64_t ldc);
int isdf_hadamard_rows(isdf_handle h, double* d_X, int64_t ldx, const double* d_Y, int64_t ldy, int rows, int64_t cols);

/* Multigrid J / LDA potential (replaces pyscf/pbc/dft/multigrid/multigrid.py:531-678 _eval_rhoG,
 * :838-935 _get_j_pass2 and the LDA branch of :1046-1150 nr_rks; host orchestration in pyscf_isdf_amd/multigrid.py).
 * Spectra are HALF spectra of real fields, (n0, n1, n2/2+1) complex128 per field, frequencies in numpy.fft.fftfreq order;
 * a level mesh may not exceed the dense mesh in any dimension.
 *   isdf_uniform_grid:          coords (3, G) of the uniform grid of ``mesh`` in cell.get_uniform_grids' order and folding
 *                               (pyscf/pbc/gto/cell.py:874-898, wrap_around = True), structure of arrays, on the device
 *   isdf_rho_pair:              rho[i, g] = sum_(mu<nA, nu<nB) aoA[mu, g] dm[i, mu, nu] aoB[nu, g]   (both AO blocks with leading
 *                               dimension ld; the rectangular form of isdf_rho: dense x (dense + sparse) pairs of a level)
 *   isdf_mg_embed_density:      spec[set] (+)= scale * fft(field[set] on mesh_sub) written at the matching frequencies of the
 *                               dense mesh (multigrid.py:665-673: tools.fft, weight, _takebak_4d); accumulate = 0 overwrites
 *                               the touched entries only - zero the spectrum first unless mesh_sub == mesh
 *   isdf_mg_restrict_potential: field[set] = scale * sum_G spec[set][G restricted to mesh_sub] e^{iGr}   (multigrid.py:862-868:
 *                               _take_4d, tools.ifft with scale = 1 / prod(mesh_sub), real part)
 *   isdf_mg_coulomb_kernel:     spec[set] *= coulG of the handle's kernel state (multigrid.py:522-525)
 *   isdf_lda_exchange:          Slater exchange of a spin-unpolarised density: exc per particle and vxc = d(rho exc)/d rho
 *                               ('lda,' of multigrid.py:1104-1106; densities <= 1e-24 give zero)
 *   isdf_lda_vwn_add:           exc += eps_c, vxc += v_c of the VWN5 correlation (libxc LDA_C_VWN: with isdf_lda_exchange the
 *                               reference's 'lda,vwn', pinned by pyscf/pbc/dft/test/test_krks.py:91-126; closed shell)
 *   isdf_gga_b88:               Becke-88 exchange ('b88,') of a spin-unpolarised density from rho and grad rho (three planes,
 *                               gstride apart): exc per particle, vrho = de/drho and w = de/d(grad rho) = 2 vsigma grad rho (three
 *                               planes, wstride apart); rho <= 1e-14 gives zero
 *   isdf_lda_exchange_fxc:      its second derivative f = d2(rho exc)/d rho2 (the LDA kernel of multigrid.py:1259-1452's response
 *                               functions; densities <= 1e-24 give zero)
 *   isdf_dot:                   *result (host) = sum x_i y_i, d_y NULL: sum x_i; fixed summation order; synchronises */
int isdf_uniform_grid(isdf_handle h, const int32_t mesh[3], const double a[9], double* d_coords_soa);
int isdf_rho_pair(isdf_handle h, const double* d_aoA, int nA, const double* d_aoB, int nB, int64_t ng, int64_t ld,
                  const double* d_dm, int nset, double* d_rho, int64_t ldrho);
int isdf_mg_embed_density(isdf_handle h, const double* d_field, int nset, const int32_t mesh_sub[3], double scale,
                          double* d_spec, const int32_t mesh[3], int accumulate);
int isdf_mg_restrict_potential(isdf_handle h, const double* d_spec, int nset, const int32_t mesh[3],
                               const int32_t mesh_sub[3], double scale, double* d_field);
int isdf_mg_coulomb_kernel(isdf_handle h, double* d_spec, int nset, const int32_t mesh[3], const double a[9]);
int isdf_lda_exchange(isdf_handle h, const double* d_rho, int64_t n, double* d_exc, double* d_vxc);
int isdf_lda_vwn_add(isdf_handle h, const double* d_rho, int64_t n, double* d_exc, double* d_vxc);
int isdf_gga_b88(isdf_handle h, const double* d_rho, const double* d_grad, int64_t gstride, int64_t n, double* d_exc,
                 double* d_vrho, double* d_w, int64_t wstride);
int isdf_lda_exchange_fxc(isdf_handle h, const double* d_rho, int64_t n, double* d_fxc);
int isdf_dot(isdf_handle h, const double* d_x, const double* d_y, int64_t n, double* result);

/* Dense helper behind S5/S6 (exposed for tests and micro-benchmarks):
 *   C (M, ldc) = alpha * A (M, lda) * (B (N, ldb) .* kscale[None, :])^T + beta * C,
 * K contiguous in both operands (the W = V Theta^T / vj = ao (v.ao)^T shape); d_kscale may be NULL.
 * Hand-written v_mfma_f64_16x16x4_f64 kernel, deterministic slab reduction. */
int isdf_gemm_nt(isdf_handle h, int M, int N, int64_t K, double alpha, const double* d_A, int64_t lda,
                 const double* d_B, int64_t ldb, const double* d_kscale, double beta, double* d_C,
                 int64_t ldc);

#ifdef __cplusplus
}
#endif
#endif

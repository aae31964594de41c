/*
 * sympgpr_hip.h -- C ABI of libsympgpr_hip.so, the MI355X (gfx950) implementation of
 * SympGPR's GP training core.  Plain C: pointers, sizes, ints; no torch / C++ types.
 *
 * This is the drop-in boundary for the ONE hot path of redmod-team/SympGPR:
 *   build_K / buildKreg            (python/05_tokamak/SympGPR/sympgpr.f90:12-60, the f2py
 *                                   object `sympgpr.build_k`, `sympgpr.buildkreg` that
 *                                   python/functions/func.py:40,50 calls)
 *   Ky = K + |sig2n| I             (python/functions/func.py:183,192)
 *   L  = cholesky(Ky, lower=True)  (python/functions/func.py:166,184,193 -> LAPACK dpotrf)
 *   alpha = L^-T L^-1 y            (python/functions/func.py:174-177       -> LAPACK dtrtrs x2)
 *   nll = y.alpha/2 + sum log L_ii (python/functions/func.py:186,195)
 *   K* rows . alpha                (sympgpr.f90:62-86,112-124: guessP / calcq / calcP target)
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - fp64, column-major (Fortran order), leading dimensions in elements.
 *   - `family`: SGPR_FAM_* selects the generated scalar-kernel file of the reference.
 *   - `hyp` / `nhyp`: the reference's hyper-parameter vector: (lx, ly, sig) for A/B/C
 *     (sympgpr.f90:17), (lx, ly, p, sig) for D
 *     (01_pendulum/implicit_period_unknown/func.py:18-19,45-46).  K is scaled by sig.
 *   - return value: 0 = ok; < 0 = SGPR_E_* (argument / HIP error, text via
 *     sgpr_last_error()); > 0 = LAPACK-style info "leading minor of order k is not positive
 *     definite" (the Python layer maps it to numpy.linalg.LinAlgError like SciPy does).
 *   - `*_host` calls take caller-owned host buffers and stage through HBM; `*_dev` calls
 *     take device pointers (hipMalloc / torch tensors) and a hipStream_t passed as void*.
 *   - There is NO CPU fallback: without a usable gfx950 device every compute call
 *     returns SGPR_E_NODEVICE.
 */
#ifndef SYMPGPR_HIP_H
#define SYMPGPR_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SGPR_ABI_VERSION 5   /* 2: solves take a writable workspace, sgpr_solve_status_dev, 12 / 7-double profile records, probes in their own library;
                                3: sgpr_fit_solve_rhs_ms, sgpr_potrf_info_dev, sgpr_trim (entry points added, none changed);
                                4: sgpr_fit_solve_rhs_dev (added);
                                5: sgpr_fit_cond_estimate, sgpr_fit_trim, sgpr_gemm_nn_dev, sgpr_trsm_rl_dev, sgpr_copy_blocks_dev (added) */

enum { SGPR_FAM_A = 0,   /* periodic(q) x SE(P), product : 05_tokamak/SympGPR/kernels.f90      */
       SGPR_FAM_B = 1,   /* periodic(q) + SE(P), sum     : 01_pendulum/explicit/kernels_sum.f90 */
       SGPR_FAM_C = 2,   /* SE x SE                      : 03_henon_heiles/kernels_sq.f90        */
       SGPR_FAM_D = 3,   /* periodic, free period p      : 01_pendulum/implicit_period_unknown/kernels.f90 */
       SGPR_FAM_USER = 4 }; /* the user's kernel: GENERATED code only (tools/gen_kernels.py, USER_FAMILY; the step the
                               reference performs with init_func.py:24-81).  As shipped: SE x SE a second time. */

/* 1 when the family's hyper-parameter vector carries a period p between the lengths and sig -- family D, or a user kernel
 * whose definition uses p: hyp = (lx, ly, p, sig), for d > 1 (lq_1..lq_d, lP_1..lP_d, p_1..p_d, sig); else 0 */
int sgpr_family_has_p(int family);

/* `which` of sgpr_kernel_eval: the four functions of a kernels*.f90 that enter K */
enum { SGPR_K_KERN = 0, SGPR_K_DXDX0 = 1, SGPR_K_DYDY0 = 2, SGPR_K_DXDY0 = 3,
       /* OR-ed in: the derivative of that function with respect to lx / ly (dkdlx_num,
        * d3kdxdx0dlx_num, ... kernels.f90:133-231; product kernels only) */
       SGPR_K_DLX = 4, SGPR_K_DLY = 8,
       /* the seven generated functions no caller of the reference uses (kernels.f90:12-57,95-132):
        * dkdx, dkdy, dkdx0, dkdy0, d3kdxdx0dy0, d3kdydy0dy0, d3kdxdy0dy0 */
       SGPR_K_DX = 16, SGPR_K_DY = 17, SGPR_K_DX0 = 18, SGPR_K_DY0 = 19, SGPR_K_DXDX0DY0 = 20,
       SGPR_K_DYDY0DY0 = 21, SGPR_K_DXDY0DY0 = 22 };

enum { SGPR_E_ARG = -1, SGPR_E_NODEVICE = -2, SGPR_E_HIP = -3, SGPR_E_NOMEM = -4,
       SGPR_E_STATE = -5 };

/* Gram-build part selection / options (sgpr_gram_pairs_dev `flags`) */
enum { SGPR_G_QQ = 1, SGPR_G_PQ = 2, SGPR_G_QP = 4, SGPR_G_PP = 8, SGPR_G_ALL = 15,
       SGPR_G_LOWER = 16,     /* write qq / PP tiles only where they touch row >= col; skip qP */
       SGPR_G_OCML = 32,      /* use the ROCm device-libs exp/sincos instead of the in-house ones */
       SGPR_G_DLX = 64,       /* entries of dK/dlx instead of K (build_dK, functions/func.py:80-129) */
       SGPR_G_DLY = 128 };    /* entries of dK/dly                                                  */

/* fit flags */
enum { SGPR_FIT_LOWER_ONLY = 1,  /* build only the lower triangle (what the factor reads)   */
       SGPR_FIT_REG = 4,         /* scalar-kernel GP (buildKreg, n = n_pts): nll_chol_reg     */
       SGPR_FIT_BLOCK_QQ = 8,    /* only the qq block of build_K (n = n_pts): nll_expl ind=0,  */
       SGPR_FIT_BLOCK_PP = 16 }; /* only the PP block, ind=1 (04_standard_map/func.py:126-141) */

int sgpr_abi_version(void);
const char *sgpr_last_error(void);
/* number of usable HIP devices (0 if none); never fails */
int sgpr_device_count(void);
/* select the device used by this thread's subsequent calls (default 0) */
int sgpr_set_device(int dev);

/* ---- stateless host-buffer calls: mirror the f2py object --------------------------------- */

/* sympgpr.build_k(x, y, x0, y0, hyp, K): K is (2n x 2n0), in/out, column-major.
 * replaces python/05_tokamak/SympGPR/sympgpr.f90:12-38 */
int sgpr_build_k_host(int family, int n, int n0, const double *x, const double *y,
                      const double *x0, const double *y0, const double *hyp, int nhyp,
                      double *K, size_t ldk);
/* sympgpr.buildkreg(x, y, x0, y0, hyp, K): K is (n x n0).  replaces sympgpr.f90:40-60 */
int sgpr_buildkreg_host(int family, int n, int n0, const double *x, const double *y,
                        const double *x0, const double *y0, const double *hyp, int nhyp,
                        double *K, size_t ldk);
/* d canonical pairs per point (BASELINE configs d = 2, 3; SURVEY.md 8 preamble): X (n x 2d), X0
 * (n0 x 2d) column-major, one column per coordinate (q_1..q_d, P_1..P_d), hyp = (lq_1..lq_d,
 * lP_1..lP_d, sig) -- (lq.., lP.., p_1..p_d, sig) for family D.  K is (2 d n x 2 d n0): block (a, b) =
 * sig d^2 k / dx_a dx'_b at rows a n, columns b n0, for the product kernel of family A (periodic q's),
 * C (all SE) or D (a free period per q), or the sum kernel B (only the diagonal blocks are non-zero).
 * d = 1 is sgpr_build_k_host entry for entry; d > 1 has no counterpart in the reference. */
int sgpr_build_k_nd_host(int family, int d, int n, int n0, const double *X, size_t ldx,
                         const double *X0, size_t ldx0, const double *hyp, int nhyp, double *K,
                         size_t ldk);
/* kernels.<name>_num, batched: out[i] = f(xa[i], ya[i], xb[i], yb[i], l...).  `l` holds
 * (lx, ly) or (lx, ly, p).  replaces the f2py `kernels` module (kernels.f90:1-94) */
int sgpr_kernel_eval_host(int family, int which, int m, const double *xa, const double *ya,
                          const double *xb, const double *yb, const double *l, int nl,
                          double *out);
/* build_dK(xin, x0in, hyp)[which] (functions/func.py:80-129; which = 0: d/dlx, 1: d/dly):
 * dK is (2 n0 x 2 n), rows index the "0" points.  Families A, C, D. */
int sgpr_build_dk_host(int family, int which, int n, int n0, const double *x, const double *y,
                       const double *x0, const double *y0, const double *hyp, int nhyp, double *dK,
                       size_t ld);
/* build_dKreg(xin, x0in, hyp)[which] (functions/func.py:52-78): dK is (n x n0) */
int sgpr_build_dkreg_host(int family, int which, int n, int n0, const double *x, const double *y,
                          const double *x0, const double *y0, const double *hyp, int nhyp, double *dK,
                          size_t ld);
/* scipy.linalg.cholesky(A, lower=True) (func.py:166): in place, strict upper zeroed. */
int sgpr_potrf_host(int n, double *A, size_t lda);
/* solve_cholesky(L, B) (func.py:174-177): B (n x nrhs) overwritten by L^-T L^-1 B. */
int sgpr_potrs_host(int n, const double *L, size_t ldl, double *B, size_t ldb, int nrhs);
/* LAPACK dsyev('V','L')-shaped: A (n x n column-major host buffer, lower triangle read) is
 * overwritten by the eigenvectors, w (n) gets the eigenvalues ascending -- the dense equivalent
 * of the drivers' `eigsh(Ky, neig)` fallback (02_pert_pendulum/func.py:199). */
int sgpr_syev_host(int n, double *A, size_t lda, double *w);

/* ---- device-resident fit: the whole nll_chol path without K ever leaving HBM --------------
 * replaces python/functions/func.py:189-196 (nll_chol) / :165-171 (gpsolve) on (x, x). */
typedef struct sgpr_fit *sgpr_fit_t;

/* Allocates HBM for an n = 2*n_pts order system (n = n_pts with SGPR_FIT_REG) and uploads
 * x, y (n_pts each), z (n). */
int sgpr_fit_create(int family, int n_pts, const double *x, const double *y, const double *z,
                    const double *hyp, int nhyp, double sig2n, unsigned flags, void *stream,
                    sgpr_fit_t *out);
/* the same fit for d canonical pairs per point: X (n_pts x 2d), z (2 d n_pts), order n = 2 d n_pts */
int sgpr_fit_create_nd(int family, int d, int n_pts, const double *X, size_t ldx, const double *z,
                       const double *hyp, int nhyp, double sig2n, unsigned flags, void *stream,
                       sgpr_fit_t *out);
/* new hyper-parameters / targets for the next run (the optimiser loop of the drivers,
 * 01_pendulum/implicit/main.py:146-151, calls the path ~100x with fixed x) */
int sgpr_fit_set_hyp(sgpr_fit_t f, const double *hyp, int nhyp, double sig2n);
int sgpr_fit_set_targets(sgpr_fit_t f, const double *z);
/* individual stages (asynchronous on the fit's stream) */
int sgpr_fit_build(sgpr_fit_t f);   /* Ky = build_K(x,x) + |sig2n| I                         */
int sgpr_fit_factor(sgpr_fit_t f);  /* L in place; returns info > 0 when not PD (this syncs)  */
int sgpr_fit_solve(sgpr_fit_t f);   /* alpha = L^-T L^-1 z ; nll                              */
/* all three; returns the factor's info */
int sgpr_fit_run(sgpr_fit_t f);
int sgpr_fit_alpha(sgpr_fit_t f, double *alpha_out /* 2*n_pts, host */);
int sgpr_fit_nll(sgpr_fit_t f, double *nll_out);
int sgpr_fit_ldiag(sgpr_fit_t f, double *diag_out /* 2*n_pts, host */);
/* copy the factor (lower, strict upper zeroed) / the matrix as built to a host buffer */
int sgpr_fit_get_matrix(sgpr_fit_t f, double *A, size_t lda);
/* extra right-hand sides with the cached factor: B (n x nrhs, host) overwritten */
int sgpr_fit_solve_rhs(sgpr_fit_t f, double *B, size_t ldb, int nrhs);
/* the same for right-hand sides that already live on the fit's device (n x nrhs, column-major, leading dimension ldb >= n,
 * overwritten with the solution): no host copies, nothing allocated after the first call -- the solve's scratch stays with the
 * fit.  Runs on the fit's stream (the caller makes sure B is complete there) and returns when the solve has finished.  This is
 * the entry a predict path that builds its right-hand sides on the device calls, and the one bench.py's --nrhs leg times. */
int sgpr_fit_solve_rhs_dev(sgpr_fit_t f, double *dB, size_t ldb, int nrhs);
/* K*(2 x 2n_pts) rows . alpha for m test points (sympgpr.f90:75-86,112-124 with alpha cached):
 * out_p[k] = Kstar(1,:).alpha, out_q[k] = Kstar(2,:).alpha */
int sgpr_fit_predict_rows(sgpr_fit_t f, int m, const double *q, const double *P, double *out_p,
                          double *out_q);
/* Ky^-1 (n x n, full symmetric, host buffer) from the cached factor: what the drivers compute with
 * scipy.linalg.inv(K + sig2n I) (01_pendulum/implicit/main.py:161) and hand to calcP / calcQ /
 * applymap as `Kyinv`.  W = L^-T by a panel solve on the identity, Ky^-1 = W W^T by the SYRK
 * kernel; two n x n scratch matrices on the device. */
int sgpr_fit_inverse(sgpr_fit_t f, double *Kyinv, size_t ld);
/* gradient of the nll with respect to (lx, ly) on a solved fit: what nll_grad / nll_grad_reg
 * return as nlp_grad (functions/func.py:132-162) */
int sgpr_fit_nll_grad(sgpr_fit_t f, double *grad2);
/* the pieces the per-example nll_grad variants recombine (03_henon_heiles/func.py:168-192,
 * 05_tokamak/SympGPR/func.py:152-168 add a third component from dK/dsig = K / sig):
 * terms5 = [alpha^T dK_lx alpha, tr(Ky^-1 dK_lx), alpha^T dK_ly alpha, tr(Ky^-1 dK_ly), tr(Ky^-1)] */
int sgpr_fit_nll_grad_terms(sgpr_fit_t f, double *terms5);
/* Eigen-decomposition of Ky = K + |sig2n| I on the device (parallel cyclic Jacobi): the
 * positive-definiteness FAILURE path of the drivers' nll_chol, which falls back to
 * `eigsh(Ky, neig, ...)` when cholesky raises (02_pert_pendulum/func.py:194-203,
 * 01_pendulum/implicit/func.py:99-114, 05_tokamak/Split_SympGPR/func.py:128-166).  Ky is rebuilt
 * (a failed factor has overwritten it) and diagonalised in place; w (n) = eigenvalues ascending,
 * c (n) = Q^T z.  The fit has to be built / run again before any other query.
 * Returns 0, or 1 if the rotations did not converge. */
int sgpr_fit_eig(sgpr_fit_t f, double *w, double *c);
/* K* . alpha for m test points with d pairs each: Xt (m x 2d), out (m x 2d), both column-major */
int sgpr_fit_predict_nd(sgpr_fit_t f, int m, const double *Xt, size_t ldxt, double *out);
/* cond_2(Ky) estimated from below with the device's own kernels (needs a valid factor): lambda_max by `iters` power iterations on
 * Ky v -- the rows of K re-evaluated from the training points by the prediction kernel, plus |sig2n| v --, lambda_min by `iters`
 * inverse iterations with the cached factor.  out4 = {lambda_max, lambda_min, cond, relative change of the quotients in the last
 * step}.  The reference never computes it; SURVEY.md 7 / 8(d) ask for it beside every parity number (the tolerance on alpha is a
 * multiple of cond * eps).  ABI 5. */
int sgpr_fit_cond_estimate(sgpr_fit_t f, int iters, double *out4);
/* milliseconds of the last build / factor / solve stage (hipEvent timing on the fit's stream) */
int sgpr_fit_stage_ms(sgpr_fit_t f, double *build_ms, double *factor_ms, double *solve_ms);
/* milliseconds of the device part of the last sgpr_fit_solve_rhs / _dev (the two triangular solves with their pack / unpack passes;
 * the host <-> device copies of B are outside): -1 when there has been none */
int sgpr_fit_solve_rhs_ms(sgpr_fit_t f, double *ms);
/* gives back the device scratch the block solves of this fit keep from call to call (sgpr_fit_solve_rhs / _dev: ~0.8 GB at
 * n = 98 304); the next block solve allocates it again.  Waits for the fit's stream.  ABI 5. */
int sgpr_fit_trim(sgpr_fit_t f);
/* device pointers of the fit (for callers that own a torch / HIP context): K/L, alpha */
int sgpr_fit_device_ptrs(sgpr_fit_t f, void **dA, size_t *lda, void **dalpha);
int sgpr_fit_destroy(sgpr_fit_t f);

/* ---- many small independent fits in ONE launch ----------------------------------------------
 * The reference's only batch axis: `nphmap` independent GP pairs (one per toroidal section) and the
 * CMA-ES populations that evaluate nll_chol for many hyper-parameter vectors
 * (python/05_tokamak/Split_SympGPR/main.py:36-41,63-66,96-112), at matrix orders 40 ... 160.
 * Problem b = 0..nbatch-1: the body of nll_chol (python/functions/func.py:189-196; with SGPR_FIT_REG
 * of nll_chol_reg, :180-187) on points x[b*n_pts ..], y[b*n_pts ..], targets z[b*n ..], hyper-parameters
 * hyp[b*nhyp ..], noise |sig2n[b]|;  n = 2 n_pts (n_pts with SGPR_FIT_REG) <= sgpr_fit_batch_max_order().
 * Outputs (host): alpha (nbatch x n, may be NULL), nll (nbatch), info (nbatch; 0 or the LAPACK-style
 * index of the first non-positive pivot of that problem -- the call itself still returns 0). */
int sgpr_fit_batch_max_order(void);
int sgpr_fit_batch(int family, int nbatch, int n_pts, const double *x, const double *y, const double *z,
                   const double *hyp, int nhyp, const double *sig2n, unsigned flags, double *alpha,
                   double *nll, int *info);

/* Gives back what the CALLING thread's earlier calls keep for re-use: the device arena and page-locked staging block of
 * sgpr_fit_batch (up to ~2 GiB after a large batch of order-2048 problems; they also shrink by themselves when a much smaller batch
 * follows) and the pooled events of its factorisations.  Never needed for correctness; call it between a hyper-parameter search
 * over small problems and a fit that wants the whole HBM.  No call of this thread may be in flight. */
int sgpr_trim(void);

/* ---- device-pointer primitives (the tiles a distributed driver composes) ------------------ */

/* Pair-tile Gram build.  For pair rows i = 0..mi-1 (row point b = (xb[i], yb[i])) and pair
 * columns j = 0..mj-1 (column point a = (xa[j], ya[j])) writes, for each selected part,
 *   qq[i + j*ld] = sig d2k/dxdx0,  Pq / qP = sig d2k/dxdy0,  PP = sig d2k/dydy0
 * (the four K(...) assignments of sympgpr.f90:27-34).  `diag_off`: global_row - global_col of
 * element (0,0); |noise| is added on qq / PP where i + diag_off == j (func.py:192). */
int sgpr_gram_pairs_dev(int family, int mi, int mj, const double *xb, const double *yb,
                        const double *xa, const double *ya, const double *hyp, int nhyp,
                        double *qq, double *Pq, double *qP, double *PP, size_t ld,
                        long diag_off, double noise, unsigned flags, void *stream);
/* scalar-kernel Gram tile: G[i + j*ld] = sig k(a_j, b_i)  (sympgpr.f90:54-59) */
int sgpr_gram_reg_dev(int family, int mi, int mj, const double *xb, const double *yb,
                      const double *xa, const double *ya, const double *hyp, int nhyp,
                      double *G, size_t ld, long diag_off, double noise, void *stream);

/* Pair-tile Gram build for d canonical pairs per point: Xb (mi x 2d), Xa (mj x 2d) column-major;
 * block (a, b) of the output at K + a*rstride + b*cstride*ld, each mi x mj. */
int sgpr_gram_nd_dev(int family, int d, int mi, int mj, const double *Xb, size_t ldxb,
                     const double *Xa, size_t ldxa, const double *hyp, int nhyp, double *K, size_t ld,
                     size_t rstride, size_t cstride, long diag_off, double noise, void *stream);
/* The same pairs, but only the blocks (a, b) with roff[a] >= 0 and coff[b] >= 0 (2d entries each, host arrays), block
 * (a, b) at K + roff[a] + coff[b] * ld: for a block-cyclic rank whose coordinate blocks hold DIFFERENT points (N / nb not a
 * multiple of the process grid), one call per pair of distinct point selections. */
int sgpr_gram_nd_sel_dev(int family, int d, int mi, int mj, const double *Xb, size_t ldxb,
                         const double *Xa, size_t ldxa, const double *hyp, int nhyp, double *K, size_t ld,
                         const long *roff, const long *coff, void *stream);
/* workspace (bytes) sgpr_potrf_dev / sgpr_trsm_rlt_dev need for order n */
size_t sgpr_potrf_workspace(int n);
/* The workspace BEGINS with the inverses of the 128 x 128 diagonal leaves of L (ceil(n / 128) blocks of 128 x 128
 * doubles): these bytes, with L itself, are all sgpr_trsm_rlt_dev / sgpr_trsv_dev read of what sgpr_potrf_dev leaves
 * there -- what a distributed driver has to send along with a diagonal block.  The rest is scratch. */
size_t sgpr_potrf_inverses_bytes(int n);
/* sgpr_potrf_dev keeps, per device, one high-priority side stream shared by all callers (every panel kernel of the device
 * runs on it, one at a time, whatever number of handles / host threads factor at once) and, for the orders the task-queue
 * driver takes (13312 .. 28672 unless SGPR_POTRF_Q=0), a pair of CU-masked streams.  They are created on first use and live until this call drains and destroys them (they come
 * back on demand).  Call it with no factorisation being enqueued on `device`; never needed for correctness -- but do call
 * it before the process ends when a profiler is attached: with the masked streams left to the runtime's own teardown a run
 * under rocprofv3 crashed in an exit handler (after its output was written).  The Python binding does so in an atexit hook.
 * Touches `device` only if this library has streams on it. */
int sgpr_release_device_streams(int device);
/* lower Cholesky in place; only the lower triangle of A is read or written.
 * dinfo: device int, 0 or the 1-based failing minor. */
int sgpr_potrf_dev(int n, double *A, size_t lda, void *work, size_t lwork, int *dinfo,
                   void *stream);
/* What the value a caller has read back from `dinfo` means: 0 and LAPACK-style positive values come back unchanged; a negative
 * value is an internal give-up (a bounded wait between the persistent kernels of the factorisation ran out: A is NOT factored,
 * restore it and call sgpr_potrf_dev again) -> SGPR_E_HIP, and the task-queue driver is switched off on `stream`'s device, so the
 * retry takes the launch-per-step driver.  (The fit handle and sgpr_potrf_host do this retry themselves.) */
int sgpr_potrf_info_dev(int info, void *stream);
/* B (m x n) := B L^-T with L (n x n) lower, its diagonal-leaf inverses in `work` as left by
 * sgpr_potrf_dev on that L (the panel solve of a right-looking step). */
int sgpr_trsm_rlt_dev(int m, int n, const double *L, size_t ldl, double *B, size_t ldb,
                      const void *work, void *stream);
/* C (m x n) := beta C + alpha A (m x k) B(n x k)^T ; lower != 0: only tiles touching
 * row >= col (+ diag_off) are computed (SYRK-style trailing update). */
int sgpr_gemm_nt_dev(int m, int n, int k, double alpha, const double *A, size_t lda,
                     const double *B, size_t ldb, double beta, double *C, size_t ldc, int lower,
                     long diag_off, void *stream);
/* SYRK-style update of a LOCAL piece of a 2-D block-cyclic matrix: as sgpr_gemm_nt_dev with
 * lower != 0, but a tile is computed iff it touches a block on/below the GLOBAL diagonal:
 * (row / blk) * pr + pi >= (col / blk) * pc + pj   (blk = block size, (pr, pc) = process grid,
 * (pi, pj) = this rank's grid coordinates plus the block offsets of C's first row / column). */
int sgpr_gemm_nt_bc_dev(int m, int n, int k, double alpha, const double *A, size_t lda,
                        const double *B, size_t ldb, double beta, double *C, size_t ldc, int blk,
                        int pr, int pi, int pc, int pj, void *stream);
/* b (n) := L^-1 b (trans = 0) or L^-T b (trans != 0); `work` as left by sgpr_potrf_dev on L.
 * The one-launch strip solves keep their tickets, progress counters and published segments IN the workspace: ONE solve
 * per workspace at a time (a second stream solving against the same factor needs its own copy of `work`). */
int sgpr_trsv_dev(int n, const double *L, size_t ldl, void *work, double *b, int trans,
                  void *stream);
/* Waits for `stream`, then: 0, or SGPR_E_HIP when a strip solve on this workspace gave up on a hand-off between two
 * workgroups (a bounded wait ran out: a device problem, never a property of the matrix).  Call it after the last
 * sgpr_trsv_dev / sgpr_potrs_vec_dev of a solve before trusting b. */
int sgpr_solve_status_dev(int n, const double *L, size_t ldl, const void *work, void *stream);
/* C (m x n) := beta C + alpha A (m x k) B (k x n)   (ABI 5; the backward half of a block of right-hand sides stored as rows) */
int sgpr_gemm_nn_dev(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B, size_t ldb,
                     double beta, double *C, size_t ldc, void *stream);
/* B (m x n) := B L^-1 with L (n x n) lower and `work` as left by sgpr_potrf_dev on it: with the right-hand sides of L^T X = Y stored
 * as the ROWS of B this is the backward solve (sgpr_trsm_rlt_dev, B := B L^-T, is the forward one).  ABI 5. */
int sgpr_trsm_rl_dev(int m, int n, const double *L, size_t ldl, double *B, size_t ldb, const void *work, void *stream);
/* cnt blocks of rows x cols doubles: block i is copied from src + i * sstep (leading dimension lds) to dst + i * dstep (ldd).
 * The packing / regrouping copies of a block-cyclic panel exchange as one launch on the caller's stream.  ABI 5. */
int sgpr_copy_blocks_dev(int rows, int cols, int cnt, const double *src, size_t lds, size_t sstep, double *dst, size_t ldd,
                         size_t dstep, void *stream);
/* y -= A x (trans = 0: A m x k, x k, y m) or y -= A^T x (trans != 0: x m, y k) */
int sgpr_gemv_sub_dev(int trans, int m, int k, const double *A, size_t lda, const double *x,
                      double *y, void *stream);
/* K*(2 x 2 n0) . alpha for m test points, everything device-resident (sympgpr.f90:75-86 calcq,
 * :112-124 target of calcP, with alpha = Kyinv ztrain cached): out_p = row 1, out_q = row 2 */
int sgpr_predict_rows_dev(int family, int m, const double *q, const double *P, int n0,
                          const double *xtrain, const double *ytrain, const double *hyp, int nhyp,
                          const double *alpha, double *out_p, double *out_q, void *stream);
/* the same for d canonical pairs: Xt (m x 2d), Xtrain (n0 x 2d), alpha (2 d n0), out (m x 2d),
 * all column-major device buffers */
int sgpr_predict_nd_dev(int family, int d, int m, const double *Xt, size_t ldxt, int n0,
                        const double *Xtrain, size_t ldxtr, const double *hyp, int nhyp,
                        const double *alpha, double *out, void *stream);
/* Kstar(1 x n0) . alpha with the scalar kernel (sympgpr.f90:62-73 guessP) */
int sgpr_predict_reg_dev(int family, int m, const double *q, const double *P, int n0,
                         const double *xtrain, const double *ytrain, const double *hyp, int nhyp,
                         const double *alpha, double *out, void *stream);
/* applymap / applymap_henon (functions/func.py:216-260; calcP / calcQ / guessP of sympgpr.f90:62-125
 * inlined) for all Ntest orbits with every time step on the device: one workgroup per orbit, the
 * implicit equation for P solved by a secant iteration from the regular-GP guess (tol 1e-13 like
 * hybrd1).  alpha = Kyinv ztrain, alphap = Kyinvp ztrainp.
 * mode bits select the per-example variants of the same recurrence:
 *   SGPR_MAP_WRAP_Q    q mod 2 pi                 (applymap; not applymap_henon, func.py:239-260)
 *   SGPR_MAP_WRAP_P    P mod 2 pi before the q update (04_standard_map/func.py:218-254)
 *   SGPR_MAP_EXPLICIT  P = p - Kstar(1,:).alpha at (q, p), no implicit solve and no first-guess GP
 *                      (01_pendulum/explicit/func_expl.py:106-128, 04_standard_map/func.py:174-179,
 *                      256-285); hypp / xtrainp / ytrainp / alphap are ignored
 *   SGPR_MAP_LOSS_NEGP P < 0 after the implicit solve ends the orbit (the tokamak maps)
 * qmap, pmap: [nm][ntest] C-ordered; a NaN marks a lost orbit from that step on.  pdiff (may be
 * NULL): the unwrapped momentum, pdiff[i+1] = pdiff[i] + (P_new - p_i) (04_standard_map/func.py:234). */
#define SGPR_MAP_WRAP_Q 1
#define SGPR_MAP_WRAP_P 2
#define SGPR_MAP_EXPLICIT 4
#define SGPR_MAP_LOSS_NEGP 8   /* implicit map: an orbit whose new momentum P is negative is lost (NaN) from that step on -- the
                                * tokamak drivers' loss test (05_tokamak/SympGPR/func.py:190-211, sympgpr.f90:128-177) without
                                * its flux-surface half (fieldlines.compute_r stays with the caller).  ABI 5. */
int sgpr_applymap_host(int family, int mode, int nm, int ntest, const double *hyp, int nhyp, int n0,
                       const double *xtrain, const double *ytrain, const double *alpha,
                       const double *hypp, int nhypp, int n0p, const double *xtrainp,
                       const double *ytrainp, const double *alphap, const double *Q0,
                       const double *P0, double *qmap, double *pmap, double *pdiff);
/* alpha-solve on device with the factor and its leaf inverses: b (n) := L^-T L^-1 b */
int sgpr_potrs_vec_dev(int n, const double *L, size_t ldl, void *work, double *b,
                       void *stream);

/* Per-launch HIP-event timing of the MFMA GEMM kernel between begin and end (measurement
 * aid for bench.py's roofline pass; not part of the reference's interface).  Events come from a
 * fixed pool (8192 launches per window; further launches are counted, not timed).
 * out12: [0..2] launches / algorithmic flop / ms of the 256x128-tile kernel for launches that had
 * the device to themselves, [3..5] all launches of the smaller tile shapes, [6..7] flop / ms of the
 * single largest launch, [8..10] 256x128-tile launches issued by the look-ahead driver (two streams
 * share the chip, their durations overlap), [11] launches beyond the pool. */
int sgpr_profile_begin(void);
int sgpr_profile_end(double *out12);
/* per-launch records of the window closed by the last sgpr_profile_end: 7 doubles each
 * (m, n, k, lower, big-tile flag, ms, overlap flag); returns the number of records available */
int sgpr_profile_launches(double *buf, int max_records);

#ifdef __cplusplus
}
#endif
#endif

// common.h -- internal declarations shared by the HIP translation units of libsympgpr_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/sympgpr_hip.h"

namespace sgpr {

void set_error(const std::string &msg);
// experiment knobs (capi.hip): built-in value unless set through libsympgpr_probe.so before first use
double tune(const char *name, double dflt);
void tune_set(const char *name, double v);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define SGPR_HIP(call)                                                          \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) return ::sgpr::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define SGPR_CHECK_LAUNCH() SGPR_HIP(hipGetLastError())

// Hyper-parameters split the way the reference does (`l = hyp[:-1]; sig = hyp[-1]`),
// plus the derived constants every Gram kernel uses.
struct KConst {
    double lx, ly, p, sig;
    double lx2, ly2;        // lx^2, ly^2
    double inv_lx2, inv_ly2;
    double cxx, cyy, cxy;   // sig*pp/lx^4, sig/ly^4, -sig*pm/(lx^2 ly^2)
    double hscale;          // 0.5 (family A/B) or p (family D); unused for C
    // length-scale derivatives (build_dK / build_dKreg)
    double inv_lx, inv_ly, inv_lx3, inv_ly3;
    double gxx;             // sig*pp: kxx = gxx (cos2h/lx^2 - sc^2/lx^4) E  (A/D),  gxx (1/lx^2 - u/lx^4) E  (C)
};
enum { DERIV_NONE = 0, DERIV_LX = 1, DERIV_LY = 2 };
int make_kconst(int family, const double *hyp, int nhyp, KConst *out);
bool family_has_p(int family);   // the family's hyp holds a period parameter p between the lengths and sig
int make_kconst_l(int family, const double *l, int nl, KConst *out);  // sig = 1

// ---- gram.hip
int gram_pairs(int family, int mi, int mj, const double *xb, const double *yb, const double *xa,
               const double *ya, const KConst &kc, double *qq, double *Pq, double *qP, double *PP,
               size_t ld, long diag_off, double noise, unsigned flags, hipStream_t st);
int gram_reg(int family, int mi, int mj, const double *xb, const double *yb, const double *xa,
             const double *ya, const KConst &kc, double *G, size_t ld, long diag_off, double noise,
             hipStream_t st, int deriv = DERIV_NONE);
int kernel_eval(int family, int which, int m, const double *xa, const double *ya, const double *xb,
                const double *yb, const KConst &kc, double *out, hipStream_t st);
int predict_rows(int family, int m, const double *q, const double *P, int n0, const double *xtr,
                 const double *ytr, const KConst &kc, const double *alpha, double *out_p,
                 double *out_q, hipStream_t st);
int predict_reg(int family, int m, const double *q, const double *P, int n0, const double *xtr,
                const double *ytr, const KConst &kc, const double *alpha, double *out,
                hipStream_t st);

int applymap_team(int ntest, int n0);          // workgroups that share one orbit
size_t applymap_team_ws(int ntest, int n0);    // bytes of device scratch applymap needs
int applymap(int family, int mode, int nm, int ntest, int n0, const double *xtr, const double *ytr,
             const KConst &kc, const double *alpha, int n0p, const double *xtrp, const double *ytrp,
             const KConst &kcp, const double *alphap, const double *Q0, const double *P0, double *qmap,
             double *pmap, double *pdiff, void *team_ws, hipStream_t st);
int applymap_status(const void *team_ws, int ntest, int n0);
unsigned applymap_last_calls();                 // K*-row evaluations (all orbits) of the last applymap of this process   // after the stream has been waited for: SGPR_E_HIP if a team gave up

// ---- gram_nd.hip : d canonical pairs per point (X: points x 2d, column-major)
int gram_nd(int family, int d, int mi, int mj, const double *Xb, size_t ldxb, const double *Xa, size_t ldxa,
            const double *hyp, int nhyp, double *K, size_t ld, size_t rstride, size_t cstride, long diag_off,
            double noise, hipStream_t st);
int gram_nd_sel(int family, int d, int mi, int mj, const double *Xb, size_t ldxb, const double *Xa, size_t ldxa,
                const double *hyp, int nhyp, double *K, size_t ld, const long *roff, const long *coff, hipStream_t st);
int predict_nd(int family, int d, int m, const double *Xt, size_t ldxt, int n0, const double *Xtr, size_t ldxtr,
               const double *hyp, int nhyp, const double *alpha, double *out, hipStream_t st);

// ---- gemm_f64.hip : C = beta C + alpha A B^T on fp64 MFMA tiles
int gemm_nt(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
            size_t ldb, double beta, double *C, size_t ldc, int lower, long diag_off,
            hipStream_t st);
int gemm_nt_bc(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
               size_t ldb, double beta, double *C, size_t ldc, int lower, const int *bc5,
               hipStream_t st);
int gemm_nn(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B, size_t ldb,
            double beta, double *C, size_t ldc, hipStream_t st);
int gemm_profile_begin();
int gemm_profile_end(double *out12);
int gemm_profile_launches(double *buf, int max_records);
void gemm_set_overlap(int on);   // look-ahead driver: launches from here on share the device (thread-local)
// diagnostics (libsympgpr_probe.so only): per-workgroup clock stamps, tile-shape / staging switches
int gemm_nt_diag(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B, size_t ldb,
                 double beta, double *C, size_t ldc, int lower, unsigned long long *stamps, int dbg, hipStream_t st);

// ---- chol.hip : leaf factor / leaf inverse / recursion / solves
constexpr int LEAF = 128;  // order of the diagonal block factored in LDS by one workgroup
// value left in *dinfo when a hand-off inside the persistent panel kernel timed out (a bug or a
// device problem, never a property of the matrix); info_status() turns it into SGPR_E_HIP
constexpr int POTRF_HANDOFF_TIMEOUT = -1000001;
constexpr int SOLVE_HANDOFF_TIMEOUT = -1000002;   // the same for the strip solves of a batch (batch.hip)
inline int info_status(int info)
{
    if (info >= 0) return info;
    set_error(info == POTRF_HANDOFF_TIMEOUT ? "potrf: a hand-off inside the panel kernel timed out"
              : info == SOLVE_HANDOFF_TIMEOUT ? "solve: a hand-off between the strips of a triangular solve timed out" : "potrf: internal error");
    return SGPR_E_HIP;
}
size_t potrf_workspace(int n);
int potrf(int n, double *A, size_t lda, void *work, size_t lwork, int *dinfo, hipStream_t st);
// batch.hip's mid-size path: nbatch factorisations of order npad (multiple of 128, <= potrf_batch_max_order()) in one launch
size_t potrf_batch_flag_bytes(int nbatch);
int potrf_batch_max_order();
int potrf_batch(int nbatch, int npad, double *A, size_t stride_a, size_t lda, double *inv, size_t stride_inv, int *flags,
                int *info, hipStream_t st);
// one panel (columns k0 .. k0 + 128 Wd, rows down to npad) of every problem; flags: potrf_batch_flag_bytes(nbatch) bytes per call
int potrf_batch_panel(int nbatch, int npad, int k0, int Wd, double *A, size_t stride_a, size_t lda, double *inv, size_t stride_inv,
                      int *flags, int *info, hipStream_t st);
bool potrf_queue_mark_failed(hipStream_t st);   // a queue factorisation gave up: the look-ahead driver from now on (true: the queue was in use)
int release_device_streams(int dev);   // drain + destroy the streams potrf created on `dev` (they come back on demand)
int trsm_rlt(int m, int n, const double *L, size_t ldl, double *B, size_t ldb, const void *work,
             hipStream_t st);
// One-right-hand-side solves keep their hand-off words (tickets, progress, the published segments) IN the workspace: one
// solve per workspace at a time (two solves against one factor from two streams need two copies of the workspace).
int potrs_vec(int n, const double *L, size_t ldl, void *work, double *b, hipStream_t st);
int solve_status(int n, const double *L, size_t ldl, const void *work, hipStream_t st);   // syncs st; SGPR_E_HIP if a strip solve gave up
int trsm_rl(int m, int n, const double *L, size_t ldl, double *B, size_t ldb, const void *work, hipStream_t st);  // B := B L^-1
int potrs_mat(int n, const double *L, size_t ldl, const void *work, double *B, size_t ldb, int nrhs, double *scratch,
              hipStream_t st);  // B (n x nrhs) := L^-T L^-1 B; scratch: potrs_mat_scratch(n, nrhs, L, ldl) bytes
size_t potrs_mat_scratch(int n, int nrhs, const double *L, size_t ldl);
bool potrs_mat_uses_strips(int n, int nrhs, const double *L, size_t ldl);   // the one-launch block solves of trsm.hip (hand-off words in `work`: solve_status)
int trsv(int n, const double *L, size_t ldl, void *work, double *b, int trans, hipStream_t st);
bool trsv_uses_strips(int n, const double *L, size_t ldl);      // potrs_vec / trsv take the one-launch strip kernels
const int *trsv_state(int n, const void *work);                 // their 8 state words: [2], [6] != 0 = a hand-off timed out
int leaf_probe(double *A, size_t lda, double *inv, int *dinfo, unsigned long long *stamps, hipStream_t st);
int leaf_inverses(int n, const double *L, size_t ldl, void *work, int *dinfo, hipStream_t st);

// ---- trsv.hip : one-right-hand-side triangular solve as one launch (strips + progress counter)
bool trsv_strips_ok(int n, const double *L, size_t ldl);
int trsv_strips(int n, const double *L, size_t ldl, const double *inv, double *b, int trans, int *state /* 4 ints, zero */,
                double *pub /* n doubles, all bytes 0xFF */, hipStream_t st);
// `nbatch` independent systems of one order in one launch: problem p at L + p sL, inv + p sInv, b / pub + p sB, state + p sState
int trsv_strips_batch(int nbatch, int n, const double *L, size_t sL, size_t ldl, const double *inv, size_t sInv, double *b, size_t sB,
                      int trans, int *state, int sState, double *pub, hipStream_t st);

// ---- trsm.hip : triangular solves with a block of right-hand sides, one launch per solve (strips + progress counter)
constexpr int TRSM_YLD = 80;                                      // row stride of the right-hand-side images [k][64 + 16]
constexpr int TRSM_FOLD = 5;                                      // tiles next to the diagonal folded into the leaf inverse
constexpr int TRSM_STATE_INTS = 16;                               // hand-off words of one forward + backward pair
bool trsm_strips_ok(int n, const double *L, size_t ldl);
void trsm_piece_of(int u, int C, int T, int out[5]);                      // the stream tickets as the host sees them (tests)
void trsm_piece_counts(int T, int C, size_t out[2]);
size_t trsm_strips_scratch(int n);                                // bytes
int potrs_strips(int n, const double *L, size_t ldl, const double *inv, double *B, size_t ldb, int nrhs, int *state /* TRSM_STATE_INTS ints */,
                 double *scratch, hipStream_t st);

// ---- batch.hip : many small fits (order <= 256 each) in one launch, one workgroup per problem
int fit_batch_max_order();
int fit_batch_trim();                            // release the calling thread's batch arena (device + pinned host)
int potrf_trim();                                // ... and its pooled events
int fit_batch(int family, int nbatch, int npts, const double *x, const double *y, const double *z, const double *hyp,
              int nhyp, const double *sig2n, unsigned flags, double *alpha, double *nll, int *info);

// ---- blas_small.hip
int zero_strict_upper(int n, double *A, size_t lda, hipStream_t st);
int sym_fill_upper(int n, double *A, size_t lda, hipStream_t st);
int nll_reduce(int n, const double *L, size_t ldl, const double *z, const double *alpha,
               double *dout /* [0]=0.5 z.alpha + sum log diag */, hipStream_t st);
int copy_diag(int n, const double *A, size_t lda, double *d, hipStream_t st);
int dot(int n, const double *a, const double *b, double *out, hipStream_t st);      // out[0] = a.b
constexpr int SUMSQ_SCRATCH = 1024;
// ---- eig.hip
int syev_jacobi(int n, double *A, size_t lda, double *V, size_t ldv, double *w_host, int max_sweeps,
                int *sweeps_done, hipStream_t st);
int sumsq(size_t count, const double *a, double *part, double *out, hipStream_t st);
int trace(int n, const double *A, size_t lda, double *out, hipStream_t st);          // out[0] = sum A_ii
int transpose(int m, int n, const double *A, size_t lda, double *B, size_t ldb, hipStream_t st);
int gemv_n_sub(int m, int k, const double *A, size_t lda, const double *x, double *y,
               hipStream_t st);  // y(m) -= A(m x k) x(k)
int gemv_t_sub(int m, int k, const double *A, size_t lda, const double *x, double *y,
               hipStream_t st);  // y(k) -= A(m x k)^T x(m)

}  // namespace sgpr

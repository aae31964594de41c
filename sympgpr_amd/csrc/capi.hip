// capi.hip -- extern "C" entry points of libsympgpr_hip.so (see include/sympgpr_hip.h).
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include <map>
#include <mutex>
#include <string>
#include "common.h"

namespace sgpr {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    g_err = std::string(hipGetErrorString(e)) + " in " + what + " (" + file + ":" + std::to_string(line) + ")";
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? SGPR_E_NOMEM : SGPR_E_HIP;
}

static int need_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device: libsympgpr_hip.so has no CPU fallback");
        return SGPR_E_NODEVICE;
    }
    return 0;
}

// Experiment knobs of the kernels and drivers (panel widths, planner constants, tile-shape switches ...): NOT environment
// variables of the product any more.  They keep their built-in values unless a measurement tool or a test sets them through
// libsympgpr_probe.so (sgpr_probe_tune) before the code that reads them runs for the first time in the process.
static std::mutex g_tune_mu;
static std::map<std::string, double> g_tune;
double tune(const char *name, double dflt)
{
    std::lock_guard<std::mutex> lock(g_tune_mu);
    const auto it = g_tune.find(name);
    return it == g_tune.end() ? dflt : it->second;
}
void tune_set(const char *name, double v)
{
    std::lock_guard<std::mutex> lock(g_tune_mu);
    g_tune[name] = v;
}

// small RAII device buffer for the host-pointer calls
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { SGPR_HIP(hipMalloc(&p, bytes ? bytes : 8)); return 0; }
    template <typename T> T *as() { return static_cast<T *>(p); }
};

static int upload(DevBuf &b, const double *h, size_t n, hipStream_t st)
{
    int rc = b.alloc(n * sizeof(double));
    if (rc) return rc;
    if (n) SGPR_HIP(hipMemcpyAsync(b.p, h, n * sizeof(double), hipMemcpyHostToDevice, st));
    return 0;
}

}  // namespace sgpr

using namespace sgpr;

struct sgpr_fit {
    int family = 0, npts = 0, n = 0;
    int d = 1;                 // canonical pairs per point (1 = the reference's layout)
    double hyp_nd[12] = {};    // (lq.., lP.., [p..,] sig) for d > 1
    int nhyp_nd = 0;
    double *dX = nullptr;      // all coordinates, (npts x 2d) column-major; dx = dX, dy = dX + npts
    unsigned flags = 0;
    hipStream_t st = nullptr;
    KConst kc{};
    double sig2n = 0.0;
    double *dx = nullptr, *dy = nullptr, *dz = nullptr, *dA = nullptr, *dalpha = nullptr;
    double *dscal = nullptr;  // [0] nll, [1] sum log diag
    int *dinfo = nullptr;
    void *work = nullptr;
    size_t lwork = 0;
    bool built = false, factored = false, solved = false;
    int info = 0;
    hipEvent_t ev[8] = {};    // build, factor, solve, solve_rhs: begin / end
    bool timed[4] = {false, false, false, false};
    void *rhs_scratch = nullptr;      // the block solves' scratch, kept from call to call (grown on demand, freed with the fit)
    size_t rhs_scratch_bytes = 0;
};

// scratch for a solve with nrhs right-hand sides: the fit's own block, grown when a call needs more.  (Allocating and freeing
// ~0.8 GB per call -- n = 98304 -- put milliseconds of idle device, a synchronising hipFree among them, in front of every
// solve; see sgpr_fit_solve_rhs_dev for what that does to the first launch behind it.)
static int rhs_scratch(sgpr_fit_t f, int nrhs, double **out)
{
    const size_t need = potrs_mat_scratch(f->n, nrhs, f->dA, (size_t)f->n);
    if (need > f->rhs_scratch_bytes) {
        if (f->rhs_scratch) { SGPR_HIP(hipStreamSynchronize(f->st)); (void)hipFree(f->rhs_scratch); }
        f->rhs_scratch = nullptr; f->rhs_scratch_bytes = 0;
        SGPR_HIP(hipMalloc(&f->rhs_scratch, need ? need : 8));
        f->rhs_scratch_bytes = need;
    }
    *out = static_cast<double *>(f->rhs_scratch);
    return 0;
}

// X = L^-T L^-1 B for a device-resident B on the fit's stream, between the events of the solve_rhs stage
static int solve_rhs_device(sgpr_fit_t f, double *dB, size_t ldb, int nrhs)
{
    int rc;
    if (nrhs >= 8 || potrs_mat_uses_strips(f->n, nrhs, f->dA, (size_t)f->n)) {
        double *dS = nullptr;
        if ((rc = rhs_scratch(f, nrhs, &dS))) return rc;
        SGPR_HIP(hipEventRecord(f->ev[6], f->st));
        if ((rc = potrs_mat(f->n, f->dA, (size_t)f->n, f->work, dB, ldb, nrhs, dS, f->st))) return rc;
        SGPR_HIP(hipEventRecord(f->ev[7], f->st));
        if ((rc = solve_status(f->n, f->dA, (size_t)f->n, f->work, f->st))) return rc;
    } else {
        SGPR_HIP(hipEventRecord(f->ev[6], f->st));
        for (int r = 0; r < nrhs; ++r) {
            if ((rc = potrs_vec(f->n, f->dA, (size_t)f->n, f->work, dB + (size_t)r * ldb, f->st))) return rc;
            if ((rc = solve_status(f->n, f->dA, (size_t)f->n, f->work, f->st))) return rc;   // the next solve reuses the hand-off words
        }
        SGPR_HIP(hipEventRecord(f->ev[7], f->st));
    }
    f->timed[3] = true;
    return 0;
}

// the strip solves bound their spins; a give-up is reported at the first call that waits for the solve
static int check_solve(sgpr_fit_t f)
{
    if (!trsv_uses_strips(f->n, f->dA, (size_t)f->n)) return 0;
    int h[8] = {};
    SGPR_HIP(hipMemcpyAsync(h, trsv_state(f->n, f->work), sizeof(h), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    if (h[2] || h[6]) { set_error("triangular solve: a hand-off between strips timed out"); return SGPR_E_HIP; }
    return 0;
}

extern "C" {

int sgpr_abi_version(void) { return SGPR_ABI_VERSION; }
const char *sgpr_last_error(void) { return g_err.c_str(); }

int sgpr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int sgpr_set_device(int dev)
{
    int rc = need_device();
    if (rc) return rc;
    SGPR_HIP(hipSetDevice(dev));
    return 0;
}

int sgpr_build_k_host(int family, int n, int n0, const double *x, const double *y, const double *x0,
                      const double *y0, const double *hyp, int nhyp, double *K, size_t ldk)
{
    int rc = need_device();
    if (rc) return rc;
    if (n < 0 || n0 < 0 || ldk < (size_t)(2 * n)) { set_error("build_k: bad shape"); return SGPR_E_ARG; }
    KConst kc;
    if ((rc = make_kconst(family, hyp, nhyp, &kc))) return rc;
    if (n == 0 || n0 == 0) return 0;
    DevBuf dx, dy, dx0, dy0, dK;
    hipStream_t st = nullptr;
    if ((rc = upload(dx, x, n, st)) || (rc = upload(dy, y, n, st)) || (rc = upload(dx0, x0, n0, st)) ||
        (rc = upload(dy0, y0, n0, st)))
        return rc;
    const size_t ld = 2 * (size_t)n;
    if ((rc = dK.alloc(ld * 2 * n0 * sizeof(double)))) return rc;
    double *k = dK.as<double>();
    rc = gram_pairs(family, n, n0, dx.as<double>(), dy.as<double>(), dx0.as<double>(), dy0.as<double>(),
                    kc, k, k + n, k + ld * n0, k + n + ld * n0, ld, 0, 0.0, SGPR_G_ALL, st);
    if (rc) return rc;
    SGPR_HIP(hipMemcpy2DAsync(K, ldk * sizeof(double), k, ld * sizeof(double), ld * sizeof(double),
                              2 * (size_t)n0, hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    return 0;
}

int sgpr_buildkreg_host(int family, int n, int n0, const double *x, const double *y, const double *x0,
                        const double *y0, const double *hyp, int nhyp, double *K, size_t ldk)
{
    int rc = need_device();
    if (rc) return rc;
    if (n < 0 || n0 < 0 || ldk < (size_t)n) { set_error("buildkreg: bad shape"); return SGPR_E_ARG; }
    KConst kc;
    if ((rc = make_kconst(family, hyp, nhyp, &kc))) return rc;
    if (n == 0 || n0 == 0) return 0;
    DevBuf dx, dy, dx0, dy0, dK;
    hipStream_t st = nullptr;
    if ((rc = upload(dx, x, n, st)) || (rc = upload(dy, y, n, st)) || (rc = upload(dx0, x0, n0, st)) ||
        (rc = upload(dy0, y0, n0, st)))
        return rc;
    const size_t ld = (size_t)n;
    if ((rc = dK.alloc(ld * n0 * sizeof(double)))) return rc;
    rc = gram_reg(family, n, n0, dx.as<double>(), dy.as<double>(), dx0.as<double>(), dy0.as<double>(), kc,
                  dK.as<double>(), ld, 0, 0.0, st);
    if (rc) return rc;
    SGPR_HIP(hipMemcpy2DAsync(K, ldk * sizeof(double), dK.p, ld * sizeof(double), ld * sizeof(double),
                              (size_t)n0, hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    return 0;
}

/* build_dK (functions/func.py:80-129), one length scale: dK is (2 n0 x 2 n), rows index the "0"
 * points; entries sig * d3k..dl(x0[k], y0[k], x[lk], y[lk]).  Every entry is even under a <-> b,
 * so the pair kernel is run with the "0" points as its row points. */
int sgpr_build_dk_host(int family, int which, int n, int n0, const double *x, const double *y,
                       const double *x0, const double *y0, const double *hyp, int nhyp, double *dK, size_t ld)
{
    int rc = need_device();
    if (rc) return rc;
    if (n < 0 || n0 < 0 || ld < (size_t)(2 * n0) || (which != 0 && which != 1)) { set_error("build_dk: bad arguments"); return SGPR_E_ARG; }
    KConst kc;
    if ((rc = make_kconst(family, hyp, nhyp, &kc))) return rc;
    if (n == 0 || n0 == 0) return 0;
    DevBuf dx, dy, dx0, dy0, dD;
    hipStream_t st = nullptr;
    if ((rc = upload(dx, x, n, st)) || (rc = upload(dy, y, n, st)) || (rc = upload(dx0, x0, n0, st)) ||
        (rc = upload(dy0, y0, n0, st)))
        return rc;
    const size_t l = 2 * (size_t)n0;
    if ((rc = dD.alloc(l * 2 * n * sizeof(double)))) return rc;
    double *d = dD.as<double>();
    rc = gram_pairs(family, n0, n, dx0.as<double>(), dy0.as<double>(), dx.as<double>(), dy.as<double>(), kc, d,
                    d + n0, d + l * n, d + n0 + l * n, l, 0, 0.0, SGPR_G_ALL | (which ? SGPR_G_DLY : SGPR_G_DLX), st);
    if (rc) return rc;
    SGPR_HIP(hipMemcpy2DAsync(dK, ld * sizeof(double), d, l * sizeof(double), l * sizeof(double), 2 * (size_t)n,
                              hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    return 0;
}

/* build_dKreg (functions/func.py:52-78): dK is (n x n0), Kp[k,lk] = sig dkdl(x0[lk], y0[lk], x[k], y[k]) */
int sgpr_build_dkreg_host(int family, int which, int n, int n0, const double *x, const double *y,
                          const double *x0, const double *y0, const double *hyp, int nhyp, double *dK, size_t ld)
{
    int rc = need_device();
    if (rc) return rc;
    if (n < 0 || n0 < 0 || ld < (size_t)n || (which != 0 && which != 1)) { set_error("build_dkreg: bad arguments"); return SGPR_E_ARG; }
    KConst kc;
    if ((rc = make_kconst(family, hyp, nhyp, &kc))) return rc;
    if (n == 0 || n0 == 0) return 0;
    DevBuf dx, dy, dx0, dy0, dD;
    hipStream_t st = nullptr;
    if ((rc = upload(dx, x, n, st)) || (rc = upload(dy, y, n, st)) || (rc = upload(dx0, x0, n0, st)) ||
        (rc = upload(dy0, y0, n0, st)))
        return rc;
    if ((rc = dD.alloc((size_t)n * n0 * sizeof(double)))) return rc;
    rc = gram_reg(family, n, n0, dx.as<double>(), dy.as<double>(), dx0.as<double>(), dy0.as<double>(), kc,
                  dD.as<double>(), (size_t)n, 0, 0.0, st, which ? DERIV_LY : DERIV_LX);
    if (rc) return rc;
    SGPR_HIP(hipMemcpy2DAsync(dK, ld * sizeof(double), dD.p, (size_t)n * sizeof(double), (size_t)n * sizeof(double),
                              (size_t)n0, hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    return 0;
}

/* d canonical pairs: X (n x 2d), X0 (n0 x 2d) column-major, hyp = (lq_1..lq_d, lP_1..lP_d, sig);
 * K (2 d n x 2 d n0), block (a, b) at rows a n, columns b n0.  d = 1 == sgpr_build_k_host. */
int sgpr_build_k_nd_host(int family, int d, int n, int n0, const double *X, size_t ldx, const double *X0,
                         size_t ldx0, const double *hyp, int nhyp, double *K, size_t ldk)
{
    int rc = need_device();
    if (rc) return rc;
    if (d < 1 || d > 3 || n < 0 || n0 < 0 || ldk < (size_t)(2 * d * n) || ldx < (size_t)n || ldx0 < (size_t)n0) {
        set_error("build_k_nd: bad shape");
        return SGPR_E_ARG;
    }
    if (n == 0 || n0 == 0) return 0;
    const int D = 2 * d;
    DevBuf dX, dX0, dK;
    hipStream_t st = nullptr;
    if ((rc = dX.alloc((size_t)n * D * sizeof(double))) || (rc = dX0.alloc((size_t)n0 * D * sizeof(double))) ||
        (rc = dK.alloc((size_t)D * n * D * n0 * sizeof(double))))
        return rc;
    SGPR_HIP(hipMemcpy2DAsync(dX.p, (size_t)n * sizeof(double), X, ldx * sizeof(double), (size_t)n * sizeof(double), D,
                              hipMemcpyHostToDevice, st));
    SGPR_HIP(hipMemcpy2DAsync(dX0.p, (size_t)n0 * sizeof(double), X0, ldx0 * sizeof(double), (size_t)n0 * sizeof(double), D,
                              hipMemcpyHostToDevice, st));
    const size_t ld = (size_t)D * n;
    if ((rc = gram_nd(family, d, n, n0, dX.as<double>(), (size_t)n, dX0.as<double>(), (size_t)n0, hyp, nhyp,
                      dK.as<double>(), ld, (size_t)n, (size_t)n0, 0, 0.0, st)))
        return rc;
    SGPR_HIP(hipMemcpy2DAsync(K, ldk * sizeof(double), dK.p, ld * sizeof(double), ld * sizeof(double), (size_t)D * n0,
                              hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    return 0;
}

int sgpr_kernel_eval_host(int family, int which, int m, const double *xa, const double *ya,
                          const double *xb, const double *yb, const double *l, int nl, double *out)
{
    int rc = need_device();
    if (rc) return rc;
    KConst kc;
    if ((rc = make_kconst_l(family, l, nl, &kc))) return rc;
    if (m <= 0) return 0;
    DevBuf a, b, c, d, o;
    hipStream_t st = nullptr;
    if ((rc = upload(a, xa, m, st)) || (rc = upload(b, ya, m, st)) || (rc = upload(c, xb, m, st)) ||
        (rc = upload(d, yb, m, st)) || (rc = o.alloc(m * sizeof(double))))
        return rc;
    rc = kernel_eval(family, which, m, a.as<double>(), b.as<double>(), c.as<double>(), d.as<double>(), kc,
                     o.as<double>(), st);
    if (rc) return rc;
    SGPR_HIP(hipMemcpyAsync(out, o.p, m * sizeof(double), hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    return 0;
}

int sgpr_potrf_host(int n, double *A, size_t lda)
{
    int rc = need_device();
    if (rc) return rc;
    if (n < 0 || (n > 0 && lda < (size_t)n)) { set_error("potrf: bad n / lda"); return SGPR_E_ARG; }
    if (n == 0) return 0;
    DevBuf dA, dW, dI;
    hipStream_t st = nullptr;
    const size_t ld = (size_t)n;
    if ((rc = dA.alloc(ld * n * sizeof(double))) || (rc = dW.alloc(potrf_workspace(n))) ||
        (rc = dI.alloc(sizeof(int))))
        return rc;
    SGPR_HIP(hipMemcpy2DAsync(dA.p, ld * sizeof(double), A, lda * sizeof(double), ld * sizeof(double), n,
                              hipMemcpyHostToDevice, st));
    int info = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if ((rc = potrf(n, dA.as<double>(), ld, dW.p, potrf_workspace(n), dI.as<int>(), st))) return rc;
        SGPR_HIP(hipMemcpyAsync(&info, dI.p, sizeof(int), hipMemcpyDeviceToHost, st));
        SGPR_HIP(hipStreamSynchronize(st));
        if (info != POTRF_HANDOFF_TIMEOUT || attempt || !potrf_queue_mark_failed(st)) break;
        // the task-queue driver gave up: the caller's matrix is still on the host -- once more, with the look-ahead driver
        SGPR_HIP(hipMemcpy2DAsync(dA.p, ld * sizeof(double), A, lda * sizeof(double), ld * sizeof(double), n,
                                  hipMemcpyHostToDevice, st));
    }
    if ((rc = zero_strict_upper(n, dA.as<double>(), ld, st))) return rc;
    SGPR_HIP(hipStreamSynchronize(st));
    if (info) return info_status(info);
    SGPR_HIP(hipMemcpy2D(A, lda * sizeof(double), dA.p, ld * sizeof(double), ld * sizeof(double), n,
                         hipMemcpyDeviceToHost));
    return 0;
}

int sgpr_potrs_host(int n, const double *L, size_t ldl, double *B, size_t ldb, int nrhs)
{
    int rc = need_device();
    if (rc) return rc;
    if (n < 0 || nrhs < 0 || (n > 0 && (ldl < (size_t)n || ldb < (size_t)n))) {
        set_error("potrs: bad shape");
        return SGPR_E_ARG;
    }
    if (n == 0 || nrhs == 0) return 0;
    DevBuf dL, dW, dB;
    hipStream_t st = nullptr;
    const size_t ld = (size_t)n;
    if ((rc = dL.alloc(ld * n * sizeof(double))) || (rc = dW.alloc(potrf_workspace(n))) ||
        (rc = dB.alloc(ld * nrhs * sizeof(double))))
        return rc;
    SGPR_HIP(hipMemcpy2DAsync(dL.p, ld * sizeof(double), L, ldl * sizeof(double), ld * sizeof(double), n,
                              hipMemcpyHostToDevice, st));
    SGPR_HIP(hipMemcpy2DAsync(dB.p, ld * sizeof(double), B, ldb * sizeof(double), ld * sizeof(double), nrhs,
                              hipMemcpyHostToDevice, st));
    if ((rc = leaf_inverses(n, dL.as<double>(), ld, dW.p, nullptr, st))) return rc;
    if (nrhs >= 8 || potrs_mat_uses_strips(n, nrhs, dL.as<double>(), ld)) {  // a block of right-hand sides on the matrix cores
        DevBuf dS;
        if ((rc = dS.alloc(potrs_mat_scratch(n, nrhs, dL.as<double>(), ld)))) return rc;
        if ((rc = potrs_mat(n, dL.as<double>(), ld, dW.p, dB.as<double>(), ld, nrhs, dS.as<double>(), st))) return rc;
        if ((rc = solve_status(n, dL.as<double>(), ld, dW.p, st))) return rc;
    } else
    for (int r = 0; r < nrhs; ++r) {
        if ((rc = potrs_vec(n, dL.as<double>(), ld, dW.p, dB.as<double>() + (size_t)r * ld, st))) return rc;
        if ((rc = solve_status(n, dL.as<double>(), ld, dW.p, st))) return rc;     // the next solve reuses the hand-off words
    }
    SGPR_HIP(hipMemcpy2DAsync(B, ldb * sizeof(double), dB.p, ld * sizeof(double), ld * sizeof(double), nrhs,
                              hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    return 0;
}

/* ---- fit handle ---------------------------------------------------------------------------- */

int sgpr_fit_destroy(sgpr_fit_t f)
{
    if (!f) return 0;
    for (void *p : {(void *)f->dX, (void *)f->dz, (void *)f->dA, (void *)f->dalpha,
                    (void *)f->dscal, (void *)f->dinfo, f->work, f->rhs_scratch})
        if (p) (void)hipFree(p);
    for (auto &e : f->ev)
        if (e) (void)hipEventDestroy(e);
    delete f;
    return 0;
}

static int fit_create_common(int family, int d, int n_pts, const double *X, size_t ldx, const double *x,
                             const double *y, const double *z, const double *hyp, int nhyp, double sig2n,
                             unsigned flags, void *stream, sgpr_fit_t *out)
{
    int rc = need_device();
    if (rc) return rc;
    if (!out || n_pts <= 0 || (d == 1 ? (!x || !y) : !X)) { set_error("fit_create: bad arguments"); return SGPR_E_ARG; }
    if (flags & ~(unsigned)(SGPR_FIT_LOWER_ONLY | SGPR_FIT_REG | SGPR_FIT_BLOCK_QQ | SGPR_FIT_BLOCK_PP)) { set_error("fit_create: unknown flag"); return SGPR_E_ARG; }
    if (d != 1 && (flags & SGPR_FIT_REG)) { set_error("fit_create: the scalar-kernel GP exists for d = 1 only"); return SGPR_E_ARG; }
    const unsigned single = flags & (SGPR_FIT_REG | SGPR_FIT_BLOCK_QQ | SGPR_FIT_BLOCK_PP);
    if ((single & (single - 1)) || (d != 1 && single)) {
        set_error("fit_create: SGPR_FIT_REG / BLOCK_QQ / BLOCK_PP are mutually exclusive and need d = 1");
        return SGPR_E_ARG;
    }
    sgpr_fit *f = new (std::nothrow) sgpr_fit;
    if (!f) return SGPR_E_NOMEM;
    f->family = family; f->npts = n_pts; f->d = d; f->flags = flags;
    f->n = single ? n_pts : 2 * d * n_pts;
    f->st = static_cast<hipStream_t>(stream);
    if (d == 1) {
        if ((rc = make_kconst(family, hyp, nhyp, &f->kc))) { delete f; return rc; }
    } else {
        const int need = family_has_p(family) ? 3 * d + 1 : 2 * d + 1;
        if (d < 1 || d > 3 || nhyp != need || !hyp || family < SGPR_FAM_A || family > SGPR_FAM_USER) {
            delete f;
            set_error("fit_create_nd: d in 1..3, hyp = (lq_1..lq_d, lP_1..lP_d, sig) -- (lq.., lP.., p_1..p_d, sig) for family D");
            return SGPR_E_ARG;
        }
        for (int i = 0; i < nhyp; ++i) f->hyp_nd[i] = hyp[i];
        f->nhyp_nd = nhyp;
    }
    f->sig2n = sig2n;
    const size_t n = (size_t)f->n;
    f->lwork = potrf_workspace(f->n);
    auto fail = [&](int code) { sgpr_fit_destroy(f); return code; };
#define FIT_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(hip_fail(e__, #call, __FILE__, __LINE__)); } while (0)
    FIT_HIP(hipMalloc((void **)&f->dX, (size_t)2 * d * n_pts * sizeof(double)));
    f->dx = f->dX;
    f->dy = f->dX + n_pts;
    FIT_HIP(hipMalloc((void **)&f->dz, n * sizeof(double)));
    FIT_HIP(hipMalloc((void **)&f->dalpha, n * sizeof(double)));
    FIT_HIP(hipMalloc((void **)&f->dscal, 4 * sizeof(double)));
    FIT_HIP(hipMalloc((void **)&f->dinfo, sizeof(int)));
    FIT_HIP(hipMalloc(&f->work, f->lwork));
    FIT_HIP(hipMalloc((void **)&f->dA, n * n * sizeof(double)));
    for (auto &e : f->ev) FIT_HIP(hipEventCreate(&e));
    if (d == 1) {
        FIT_HIP(hipMemcpyAsync(f->dx, x, n_pts * sizeof(double), hipMemcpyHostToDevice, f->st));
        FIT_HIP(hipMemcpyAsync(f->dy, y, n_pts * sizeof(double), hipMemcpyHostToDevice, f->st));
    } else {
        FIT_HIP(hipMemcpy2DAsync(f->dX, n_pts * sizeof(double), X, ldx * sizeof(double), n_pts * sizeof(double),
                                 (size_t)2 * d, hipMemcpyHostToDevice, f->st));
    }
    if (z) FIT_HIP(hipMemcpyAsync(f->dz, z, n * sizeof(double), hipMemcpyHostToDevice, f->st));
    else FIT_HIP(hipMemsetAsync(f->dz, 0, n * sizeof(double), f->st));
    FIT_HIP(hipStreamSynchronize(f->st));
#undef FIT_HIP
    *out = f;
    return 0;
}

int sgpr_fit_create(int family, int n_pts, const double *x, const double *y, const double *z,
                    const double *hyp, int nhyp, double sig2n, unsigned flags, void *stream,
                    sgpr_fit_t *out)
{
    return fit_create_common(family, 1, n_pts, nullptr, 0, x, y, z, hyp, nhyp, sig2n, flags, stream, out);
}

int sgpr_fit_create_nd(int family, int d, int n_pts, const double *X, size_t ldx, const double *z,
                       const double *hyp, int nhyp, double sig2n, unsigned flags, void *stream,
                       sgpr_fit_t *out)
{
    if (d == 1) {
        if (!X || ldx < (size_t)n_pts) { set_error("fit_create_nd: bad X"); return SGPR_E_ARG; }
        return fit_create_common(family, 1, n_pts, nullptr, 0, X, X + ldx, z, hyp, nhyp, sig2n, flags, stream, out);
    }
    if (!X || ldx < (size_t)(n_pts > 0 ? n_pts : 1)) { set_error("fit_create_nd: bad X"); return SGPR_E_ARG; }
    return fit_create_common(family, d, n_pts, X, ldx, nullptr, nullptr, z, hyp, nhyp, sig2n, flags, stream, out);
}

int sgpr_fit_set_hyp(sgpr_fit_t f, const double *hyp, int nhyp, double sig2n)
{
    if (!f) { set_error("null fit"); return SGPR_E_ARG; }
    if (f->d > 1) {
        if (!hyp || nhyp != f->nhyp_nd) { set_error("fit_set_hyp: hyp = (lq.., lP.., [p..,] sig)"); return SGPR_E_ARG; }
        for (int i = 0; i < nhyp; ++i) f->hyp_nd[i] = hyp[i];
    } else {
        int rc = make_kconst(f->family, hyp, nhyp, &f->kc);
        if (rc) return rc;
    }
    f->sig2n = sig2n;
    f->built = f->factored = f->solved = false;
    return 0;
}

int sgpr_fit_set_targets(sgpr_fit_t f, const double *z)
{
    if (!f || !z) { set_error("null argument"); return SGPR_E_ARG; }
    SGPR_HIP(hipMemcpyAsync(f->dz, z, (size_t)f->n * sizeof(double), hipMemcpyHostToDevice, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    f->solved = false;
    return 0;
}

static int fit_build_impl(sgpr_fit_t f, bool lower_only)
{
    if (!f) { set_error("null fit"); return SGPR_E_ARG; }
    const size_t n = (size_t)f->n;
    const int N = f->npts;
    unsigned flags = SGPR_G_ALL;
    if (lower_only) flags |= SGPR_G_LOWER;
    SGPR_HIP(hipEventRecord(f->ev[0], f->st));
    int rc;
    if (f->d > 1) {
        // d canonical pairs: (2d)^2 blocks of N x N, block (a, b) at rows a N, columns b N
        rc = gram_nd(f->family, f->d, N, N, f->dX, (size_t)N, f->dX, (size_t)N, f->hyp_nd, f->nhyp_nd, f->dA, n,
                     (size_t)N, (size_t)N, 0, std::fabs(f->sig2n), f->st);
    } else if (f->flags & SGPR_FIT_REG) {
        // Ky = buildKreg(x, x) + |sig2n| I  (func.py:182-183)
        rc = gram_reg(f->family, N, N, f->dx, f->dy, f->dx, f->dy, f->kc, f->dA, n, 0, std::fabs(f->sig2n), f->st);
    } else if (f->flags & (SGPR_FIT_BLOCK_QQ | SGPR_FIT_BLOCK_PP)) {
        // one diagonal block of build_K(x, x) + |sig2n| I  (04_standard_map/func.py:126-135)
        const unsigned part = (f->flags & SGPR_FIT_BLOCK_QQ) ? SGPR_G_QQ : SGPR_G_PP;
        rc = gram_pairs(f->family, N, N, f->dx, f->dy, f->dx, f->dy, f->kc, f->dA, f->dA, f->dA, f->dA, n, 0,
                        std::fabs(f->sig2n), part | (flags & SGPR_G_LOWER), f->st);
    } else
    // Ky = build_K(x, x) + |sig2n| I  (func.py:191-192), noise fused into the diagonal tiles
    rc = gram_pairs(f->family, N, N, f->dx, f->dy, f->dx, f->dy, f->kc, f->dA, f->dA + N,
                        f->dA + n * N, f->dA + N + n * N, n, 0, std::fabs(f->sig2n), flags, f->st);
    if (rc) return rc;
    SGPR_HIP(hipEventRecord(f->ev[1], f->st));
    f->timed[0] = true;
    f->built = true;
    f->factored = f->solved = false;
    return 0;
}

int sgpr_fit_build(sgpr_fit_t f) { return fit_build_impl(f, f && (f->flags & SGPR_FIT_LOWER_ONLY)); }

/* Eigen-decomposition of Ky = K + |sig2n| I on the device (parallel cyclic Jacobi, eig.hip): the
 * positive-definiteness failure path of the drivers' nll_chol, which falls back to
 * `eigsh(Ky, neig, ...)` when cholesky raises (02_pert_pendulum/func.py:194-203).  Ky is rebuilt
 * (a failed factorisation has overwritten it), diagonalised in place, and
 * w (n, ascending eigenvalues) and c = Q^T z (n) come back; the caller forms
 * alpha = Q diag(1/w) c and the log-determinant from whichever eigenpairs it keeps.
 * Two n x n matrices in HBM.  Returns 0, or 1 if the rotations did not converge in 40 sweeps. */
int sgpr_fit_eig(sgpr_fit_t f, double *w, double *c)
{
    if (!f || !w || !c) { set_error("null argument"); return SGPR_E_ARG; }
    int rc = fit_build_impl(f, false);
    if (rc) return rc;
    const size_t n = (size_t)f->n;
    DevBuf V, tmp;
    if ((rc = V.alloc(n * n * sizeof(double))) || (rc = tmp.alloc(n * sizeof(double)))) return rc;
    int sweeps = 0;
    const int st = syev_jacobi(f->n, f->dA, n, V.as<double>(), n, w, 40, &sweeps, f->st);
    f->built = f->factored = f->solved = false;   // dA now holds the eigenvectors
    if (st < 0) return st;
    SGPR_HIP(hipMemsetAsync(tmp.p, 0, n * sizeof(double), f->st));
    if ((rc = gemv_t_sub(f->n, f->n, f->dA, n, f->dz, tmp.as<double>(), f->st))) return rc;   // tmp = -Q^T z
    SGPR_HIP(hipMemcpyAsync(c, tmp.p, n * sizeof(double), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    for (size_t i = 0; i < n; ++i) c[i] = -c[i];
    if (st > 0) set_error("fit_eig: Jacobi sweeps did not converge");
    return st;
}

/* LAPACK dsyev('V', 'L') shaped host call: A (n x n, column-major, lower triangle read) is
 * overwritten by the eigenvectors, w (n) receives the eigenvalues in ascending order. */
int sgpr_syev_host(int n, double *A, size_t lda, double *w)
{
    int rc = need_device();
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!A || !w || lda < (size_t)n))) { set_error("syev: bad arguments"); return SGPR_E_ARG; }
    if (n == 0) return 0;
    const size_t N = (size_t)n;
    DevBuf dA, dV;
    if ((rc = dA.alloc(N * N * sizeof(double))) || (rc = dV.alloc(N * N * sizeof(double)))) return rc;
    hipStream_t st = nullptr;
    SGPR_HIP(hipMemcpy2DAsync(dA.p, N * sizeof(double), A, lda * sizeof(double), N * sizeof(double), N,
                              hipMemcpyHostToDevice, st));
    if ((rc = sym_fill_upper(n, dA.as<double>(), N, st))) return rc;
    int sweeps = 0;
    const int status = syev_jacobi(n, dA.as<double>(), N, dV.as<double>(), N, w, 40, &sweeps, st);
    if (status < 0) return status;
    SGPR_HIP(hipMemcpy2DAsync(A, lda * sizeof(double), dA.p, N * sizeof(double), N * sizeof(double), N,
                              hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    if (status > 0) set_error("syev: Jacobi sweeps did not converge");
    return status;
}

int sgpr_fit_factor(sgpr_fit_t f)
{
    if (!f) { set_error("null fit"); return SGPR_E_ARG; }
    if (!f->built) { set_error("fit_factor: call sgpr_fit_build first"); return SGPR_E_STATE; }
    SGPR_HIP(hipEventRecord(f->ev[2], f->st));
    int rc = potrf(f->n, f->dA, (size_t)f->n, f->work, f->lwork, f->dinfo, f->st);
    if (rc) return rc;
    SGPR_HIP(hipEventRecord(f->ev[3], f->st));
    f->timed[1] = true;
    f->built = false;  // K has been overwritten by L
    SGPR_HIP(hipMemcpyAsync(&f->info, f->dinfo, sizeof(int), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    if (f->info == POTRF_HANDOFF_TIMEOUT && potrf_queue_mark_failed(f->st)) {
        // The task-queue driver gave up (a hand-off between its persistent kernels ran into its time limit): Ky is half
        // overwritten, but the handle holds what it was built from.  Build it again and factor with the look-ahead driver,
        // which this device uses from now on.  (Once: a second give-up is an error.)
        if ((rc = fit_build_impl(f, f->flags & SGPR_FIT_LOWER_ONLY))) return rc;
        SGPR_HIP(hipEventRecord(f->ev[2], f->st));
        if ((rc = potrf(f->n, f->dA, (size_t)f->n, f->work, f->lwork, f->dinfo, f->st))) return rc;
        SGPR_HIP(hipEventRecord(f->ev[3], f->st));
        f->built = false;
        SGPR_HIP(hipMemcpyAsync(&f->info, f->dinfo, sizeof(int), hipMemcpyDeviceToHost, f->st));
        SGPR_HIP(hipStreamSynchronize(f->st));
    }
    f->factored = f->info == 0;
    return info_status(f->info);
}

int sgpr_fit_solve(sgpr_fit_t f)
{
    if (!f) { set_error("null fit"); return SGPR_E_ARG; }
    if (!f->factored) { set_error("fit_solve: no valid factor"); return SGPR_E_STATE; }
    const size_t n = (size_t)f->n;
    SGPR_HIP(hipEventRecord(f->ev[4], f->st));
    SGPR_HIP(hipMemcpyAsync(f->dalpha, f->dz, n * sizeof(double), hipMemcpyDeviceToDevice, f->st));
    int rc = potrs_vec(f->n, f->dA, n, f->work, f->dalpha, f->st);
    if (rc) return rc;
    if ((rc = nll_reduce(f->n, f->dA, n, f->dz, f->dalpha, f->dscal, f->st))) return rc;
    SGPR_HIP(hipEventRecord(f->ev[5], f->st));
    f->timed[2] = true;
    f->solved = true;
    return 0;
}

int sgpr_fit_run(sgpr_fit_t f)
{
    int rc = sgpr_fit_build(f);
    if (rc) return rc;
    if ((rc = sgpr_fit_factor(f))) return rc;
    return sgpr_fit_solve(f);
}

int sgpr_fit_alpha(sgpr_fit_t f, double *alpha_out)
{
    if (!f || !alpha_out) { set_error("null argument"); return SGPR_E_ARG; }
    if (!f->solved) { set_error("fit_alpha: not solved"); return SGPR_E_STATE; }
    SGPR_HIP(hipMemcpyAsync(alpha_out, f->dalpha, (size_t)f->n * sizeof(double), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    return check_solve(f);
}

int sgpr_fit_nll(sgpr_fit_t f, double *nll_out)
{
    if (!f || !nll_out) { set_error("null argument"); return SGPR_E_ARG; }
    if (!f->solved) { set_error("fit_nll: not solved"); return SGPR_E_STATE; }
    SGPR_HIP(hipMemcpyAsync(nll_out, f->dscal, sizeof(double), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    return check_solve(f);
}

int sgpr_fit_ldiag(sgpr_fit_t f, double *diag_out)
{
    if (!f || !diag_out) { set_error("null argument"); return SGPR_E_ARG; }
    if (!f->factored) { set_error("fit_ldiag: no valid factor"); return SGPR_E_STATE; }
    DevBuf d;
    int rc = d.alloc((size_t)f->n * sizeof(double));
    if (rc) return rc;
    if ((rc = copy_diag(f->n, f->dA, (size_t)f->n, d.as<double>(), f->st))) return rc;
    SGPR_HIP(hipMemcpyAsync(diag_out, d.p, (size_t)f->n * sizeof(double), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    return 0;
}

int sgpr_fit_get_matrix(sgpr_fit_t f, double *A, size_t lda)
{
    if (!f || !A || lda < (size_t)f->n) { set_error("fit_get_matrix: bad arguments"); return SGPR_E_ARG; }
    if (!f->factored && !f->built) { set_error("fit_get_matrix: nothing built"); return SGPR_E_STATE; }
    const size_t n = (size_t)f->n;
    int rc;
    if (f->factored) {
        if ((rc = zero_strict_upper(f->n, f->dA, n, f->st))) return rc;
    } else if (f->flags & SGPR_FIT_LOWER_ONLY) {
        if ((rc = sym_fill_upper(f->n, f->dA, n, f->st))) return rc;
    }
    SGPR_HIP(hipMemcpy2DAsync(A, lda * sizeof(double), f->dA, n * sizeof(double), n * sizeof(double), n,
                              hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    return 0;
}

int sgpr_fit_solve_rhs(sgpr_fit_t f, double *B, size_t ldb, int nrhs)
{
    if (!f || !B || ldb < (size_t)f->n || nrhs < 0) { set_error("fit_solve_rhs: bad arguments"); return SGPR_E_ARG; }
    if (!f->factored) { set_error("fit_solve_rhs: no valid factor"); return SGPR_E_STATE; }
    if (nrhs == 0) return 0;
    const size_t n = (size_t)f->n;
    DevBuf dB;
    int rc = dB.alloc(n * nrhs * sizeof(double));
    if (rc) return rc;
    SGPR_HIP(hipMemcpy2DAsync(dB.p, n * sizeof(double), B, ldb * sizeof(double), n * sizeof(double), nrhs,
                              hipMemcpyHostToDevice, f->st));
    if ((rc = solve_rhs_device(f, dB.as<double>(), n, nrhs))) return rc;
    SGPR_HIP(hipMemcpy2DAsync(B, ldb * sizeof(double), dB.p, n * sizeof(double), n * sizeof(double), nrhs,
                              hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    return 0;
}

int sgpr_fit_solve_rhs_dev(sgpr_fit_t f, double *dB, size_t ldb, int nrhs)
{
    if (!f || !dB || ldb < (size_t)f->n || nrhs < 0 || (((uintptr_t)dB) & 7)) { set_error("fit_solve_rhs_dev: bad arguments"); return SGPR_E_ARG; }
    if (!f->factored) { set_error("fit_solve_rhs_dev: no valid factor"); return SGPR_E_STATE; }
    if (nrhs == 0) return 0;
    return solve_rhs_device(f, dB, ldb, nrhs);
}

int sgpr_fit_predict_rows(sgpr_fit_t f, int m, const double *q, const double *P, double *out_p,
                          double *out_q)
{
    if (!f || m < 0 || !q || !P || !out_p || !out_q) { set_error("fit_predict_rows: bad arguments"); return SGPR_E_ARG; }
    if (!f->solved) { set_error("fit_predict_rows: not solved"); return SGPR_E_STATE; }
    if (f->d > 1) { set_error("fit_predict_rows: use sgpr_fit_predict_nd for d > 1"); return SGPR_E_STATE; }
    if (f->flags & (SGPR_FIT_BLOCK_QQ | SGPR_FIT_BLOCK_PP)) { set_error("fit_predict_rows: not defined for a single-block fit"); return SGPR_E_STATE; }
    if (m == 0) return 0;
    DevBuf dq, dP, dop, doq;
    int rc;
    if ((rc = upload(dq, q, m, f->st)) || (rc = upload(dP, P, m, f->st)) ||
        (rc = dop.alloc(m * sizeof(double))) || (rc = doq.alloc(m * sizeof(double))))
        return rc;
    SGPR_HIP(hipMemsetAsync(doq.p, 0, m * sizeof(double), f->st));
    if (f->flags & SGPR_FIT_REG)  // scalar-kernel GP: one row per test point, written to out_p; out_q = 0
        rc = predict_reg(f->family, m, dq.as<double>(), dP.as<double>(), f->npts, f->dx, f->dy, f->kc,
                         f->dalpha, dop.as<double>(), f->st);
    else
    rc = predict_rows(f->family, m, dq.as<double>(), dP.as<double>(), f->npts, f->dx, f->dy, f->kc,
                      f->dalpha, dop.as<double>(), doq.as<double>(), f->st);
    if (rc) return rc;
    SGPR_HIP(hipMemcpyAsync(out_p, dop.p, m * sizeof(double), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipMemcpyAsync(out_q, doq.p, m * sizeof(double), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    return 0;
}

int sgpr_fit_inverse(sgpr_fit_t f, double *Kyinv, size_t ld)
{
    if (!f || !Kyinv || ld < (size_t)f->n) { set_error("fit_inverse: bad arguments"); return SGPR_E_ARG; }
    if (!f->factored) { set_error("fit_inverse: no valid factor"); return SGPR_E_STATE; }
    const size_t n = (size_t)f->n;
    DevBuf W, R;
    int rc;
    if ((rc = W.alloc(n * n * sizeof(double))) || (rc = R.alloc(n * n * sizeof(double)))) return rc;
    double *w = W.as<double>(), *r = R.as<double>();
    SGPR_HIP(hipMemsetAsync(w, 0, n * n * sizeof(double), f->st));
    {   // identity: one strided memset-like copy of ones onto the diagonal
        std::vector<double> ones(n, 1.0);
        SGPR_HIP(hipMemcpy2DAsync(w, (n + 1) * sizeof(double), ones.data(), sizeof(double), sizeof(double), n,
                                  hipMemcpyHostToDevice, f->st));
        SGPR_HIP(hipStreamSynchronize(f->st));
    }
    if ((rc = trsm_rlt(f->n, f->n, f->dA, n, w, n, f->work, f->st))) return rc;            // W = L^-T
    if ((rc = gemm_nt(f->n, f->n, f->n, 1.0, w, n, w, n, 0.0, r, n, 1, 0, f->st))) return rc;  // lower(W W^T)
    if ((rc = sym_fill_upper(f->n, r, n, f->st))) return rc;
    SGPR_HIP(hipMemcpy2DAsync(Kyinv, ld * sizeof(double), r, n * sizeof(double), n * sizeof(double), n,
                              hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    return 0;
}

/* d nll / d(lx, ly) as nll_grad / nll_grad_reg compute it (functions/func.py:132-162):
 *   grad_i = -1/2 alpha^T dK_i alpha + 1/2 tr(Ky^-1 dK_i).
 * The reference forms Ky^-1 explicitly; here tr(Ky^-1 dK) = tr(L^-1 dK L^-T): W = dK, W := W L^-T
 * (panel solve), W := W^T (= L^-1 dK by symmetry), W := W L^-T again, sum of the diagonal --
 * 2 n^3 flop per length scale on the MFMA kernel, two n x n scratch matrices. */
static int nll_grad_core(sgpr_fit_t f, double *h)
{
    if (!f || !h) { set_error("null argument"); return SGPR_E_ARG; }
    if (!f->solved) { set_error("fit_nll_grad: run the fit first"); return SGPR_E_STATE; }
    if (f->d > 1) { set_error("fit_nll_grad: available for d = 1"); return SGPR_E_STATE; }
    if (f->flags & (SGPR_FIT_BLOCK_QQ | SGPR_FIT_BLOCK_PP)) { set_error("fit_nll_grad: not defined for a single-block fit"); return SGPR_E_STATE; }
    const size_t n = (size_t)f->n;
    const int N = f->npts;
    DevBuf W, T, tmp, sc;
    int rc;
    if ((rc = W.alloc(n * n * sizeof(double))) || (rc = T.alloc(n * n * sizeof(double))) ||
        (rc = tmp.alloc(n * sizeof(double))) || (rc = sc.alloc(4 * sizeof(double))))
        return rc;
    double *w = W.as<double>(), *t = T.as<double>(), *s = sc.as<double>();
    for (int which = 0; which < 2; ++which) {
        if (f->flags & SGPR_FIT_REG)
            rc = gram_reg(f->family, N, N, f->dx, f->dy, f->dx, f->dy, f->kc, w, n, 0, 0.0, f->st,
                          which ? DERIV_LY : DERIV_LX);
        else
            rc = gram_pairs(f->family, N, N, f->dx, f->dy, f->dx, f->dy, f->kc, w, w + N, w + n * N, w + N + n * N,
                            n, 0, 0.0, SGPR_G_ALL | (which ? SGPR_G_DLY : SGPR_G_DLX), f->st);
        if (rc) return rc;
        // alpha^T dK alpha
        SGPR_HIP(hipMemsetAsync(tmp.p, 0, n * sizeof(double), f->st));
        if ((rc = gemv_n_sub(f->n, f->n, w, n, f->dalpha, tmp.as<double>(), f->st))) return rc;  // tmp = -dK alpha
        if ((rc = dot(f->n, tmp.as<double>(), f->dalpha, s + 2 * which, f->st))) return rc;
        // tr(L^-1 dK L^-T)
        if ((rc = trsm_rlt(f->n, f->n, f->dA, n, w, n, f->work, f->st))) return rc;
        if ((rc = transpose(f->n, f->n, w, n, t, n, f->st))) return rc;
        if ((rc = trsm_rlt(f->n, f->n, f->dA, n, t, n, f->work, f->st))) return rc;
        if ((rc = trace(f->n, t, n, s + 2 * which + 1, f->st))) return rc;
    }
    SGPR_HIP(hipMemcpyAsync(h, s, 4 * sizeof(double), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    h[0] = -h[0];   // the GEMV helper subtracts: s[0], s[2] hold -(alpha^T dK alpha)
    h[2] = -h[2];
    return 0;
}

int sgpr_fit_nll_grad(sgpr_fit_t f, double *grad2)
{
    if (!grad2) { set_error("null argument"); return SGPR_E_ARG; }
    double h[4];
    int rc = nll_grad_core(f, h);
    if (rc) return rc;
    for (int which = 0; which < 2; ++which) grad2[which] = -0.5 * h[2 * which] + 0.5 * h[2 * which + 1];
    return 0;
}

/* The pieces the per-example nll_grad variants recombine (03_henon_heiles/func.py:168-192,
 * 05_tokamak/SympGPR/func.py:152-168: a third component built from dK/dsig = K / sig):
 * terms5 = [alpha^T dK_lx alpha, tr(Ky^-1 dK_lx), alpha^T dK_ly alpha, tr(Ky^-1 dK_ly), tr(Ky^-1)].
 * tr(Ky^-1) = ||L^-1||_F^2 from a panel solve on the identity. */
int sgpr_fit_nll_grad_terms(sgpr_fit_t f, double *terms5)
{
    if (!terms5) { set_error("null argument"); return SGPR_E_ARG; }
    int rc = nll_grad_core(f, terms5);
    if (rc) return rc;
    const size_t n = (size_t)f->n;
    DevBuf W, sc;
    if ((rc = W.alloc(n * n * sizeof(double))) || (rc = sc.alloc((SUMSQ_SCRATCH + 1) * sizeof(double)))) return rc;
    double *w = W.as<double>();
    SGPR_HIP(hipMemsetAsync(w, 0, n * n * sizeof(double), f->st));
    {
        std::vector<double> ones(n, 1.0);
        SGPR_HIP(hipMemcpy2DAsync(w, (n + 1) * sizeof(double), ones.data(), sizeof(double), sizeof(double), n,
                                  hipMemcpyHostToDevice, f->st));
        SGPR_HIP(hipStreamSynchronize(f->st));
    }
    if ((rc = trsm_rlt(f->n, f->n, f->dA, n, w, n, f->work, f->st))) return rc;            // W = L^-T
    if ((rc = sumsq(n * n, w, sc.as<double>() + 1, sc.as<double>(), f->st))) return rc;
    SGPR_HIP(hipMemcpyAsync(terms5 + 4, sc.p, sizeof(double), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    return 0;
}

/* K*(2d x 2d N) . alpha for m test points Xt (m x 2d, column-major, leading dimension ldxt):
 * out (m x 2d, column-major, ld m): column a = predicted d F / d x_a */
int sgpr_fit_predict_nd(sgpr_fit_t f, int m, const double *Xt, size_t ldxt, double *out)
{
    if (!f || m < 0 || !Xt || !out || ldxt < (size_t)(m > 0 ? m : 1)) { set_error("fit_predict_nd: bad arguments"); return SGPR_E_ARG; }
    if (!f->solved) { set_error("fit_predict_nd: not solved"); return SGPR_E_STATE; }
    if (m == 0) return 0;
    const int D = 2 * f->d;
    DevBuf dT, dO;
    int rc;
    if ((rc = dT.alloc((size_t)m * D * sizeof(double))) || (rc = dO.alloc((size_t)m * D * sizeof(double)))) return rc;
    SGPR_HIP(hipMemcpy2DAsync(dT.p, (size_t)m * sizeof(double), Xt, ldxt * sizeof(double), (size_t)m * sizeof(double), D,
                              hipMemcpyHostToDevice, f->st));
    double hyp1[4] = {f->kc.lx, f->kc.ly, f->kc.sig, 0.0};
    int nh1 = 3;
    if (family_has_p(f->family)) { hyp1[2] = f->kc.p; hyp1[3] = f->kc.sig; nh1 = 4; }
    const double *hyp = f->d > 1 ? f->hyp_nd : hyp1;
    if ((rc = predict_nd(f->family, f->d, m, dT.as<double>(), (size_t)m, f->npts, f->dX, (size_t)f->npts, hyp,
                         f->d > 1 ? f->nhyp_nd : nh1,
                         f->dalpha, dO.as<double>(), f->st)))
        return rc;
    SGPR_HIP(hipMemcpyAsync(out, dO.p, (size_t)m * D * sizeof(double), hipMemcpyDeviceToHost, f->st));
    SGPR_HIP(hipStreamSynchronize(f->st));
    return 0;
}

// cond_2(Ky) from below: lambda_max by power iteration on Ky v (the rows of K are re-evaluated from the training points by the
// prediction kernel -- the matrix itself has been overwritten by its factor -- plus |sig2n| v), lambda_min by inverse iteration
// with the cached factor (two strip solves per step).  Both are Rayleigh quotients of unit vectors, so lambda_max is a lower
// and lambda_min an upper bound: the estimate never exceeds the true condition number.  Vector arithmetic on the host (n
// doubles per step); out4 = {lambda_max, lambda_min, cond, relative change of the two quotients in their last step (the larger)}.
// "report cond (or a Lanczos estimate) next to every parity number".
int sgpr_fit_cond_estimate(sgpr_fit_t f, int iters, double *out4)
{
    if (!f || !out4 || iters < 1) { set_error("fit_cond_estimate: bad arguments"); return SGPR_E_ARG; }
    if (!f->factored) { set_error("fit_cond_estimate: no valid factor"); return SGPR_E_STATE; }
    if (f->flags & (SGPR_FIT_BLOCK_QQ | SGPR_FIT_BLOCK_PP)) { set_error("fit_cond_estimate: not defined for a single-block fit"); return SGPR_E_STATE; }
    const size_t n = (size_t)f->n;
    const int m = f->npts;
    DevBuf dv, dw;
    int rc;
    if ((rc = dv.alloc(n * sizeof(double))) || (rc = dw.alloc(n * sizeof(double)))) return rc;
    std::vector<double> v(n), w(n);
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    double nrm = 0.0;
    for (size_t i = 0; i < n; ++i) {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        v[i] = (double)(lcg >> 11) / 9007199254740992.0 - 0.5;
        nrm += v[i] * v[i];
    }
    nrm = std::sqrt(nrm);
    for (size_t i = 0; i < n; ++i) v[i] /= nrm;
    const std::vector<double> v0 = v;
    double hyp1[4] = {f->kc.lx, f->kc.ly, f->kc.sig, 0.0};
    int nh1 = 3;
    if (family_has_p(f->family)) { hyp1[2] = f->kc.p; hyp1[3] = f->kc.sig; nh1 = 4; }
    auto normalise = [&](const std::vector<double> &src, std::vector<double> &dst, double &quot) {
        double dot = 0.0, nn = 0.0;
        for (size_t i = 0; i < n; ++i) { dot += dst[i] * src[i]; nn += src[i] * src[i]; }
        quot = dot;                                     // v^T (M v), |v| = 1
        nn = std::sqrt(nn);
        for (size_t i = 0; i < n; ++i) dst[i] = src[i] / nn;
    };
    double lmax = 0.0, lmin_inv = 0.0, ch_max = 1.0, ch_min = 1.0;
    for (int it = 0; it < iters; ++it) {                // ---- lambda_max
        SGPR_HIP(hipMemcpyAsync(dv.p, v.data(), n * sizeof(double), hipMemcpyHostToDevice, f->st));
        if (f->flags & SGPR_FIT_REG) {
            rc = predict_reg(f->family, m, f->dx, f->dy, m, f->dx, f->dy, f->kc, dv.as<double>(), dw.as<double>(), f->st);
        } else if (f->d > 1) {
            rc = predict_nd(f->family, f->d, m, f->dX, (size_t)m, m, f->dX, (size_t)m, f->hyp_nd, f->nhyp_nd, dv.as<double>(),
                            dw.as<double>(), f->st);
        } else {
            rc = predict_rows(f->family, m, f->dx, f->dy, m, f->dx, f->dy, f->kc, dv.as<double>(), dw.as<double>(),
                              dw.as<double>() + m, f->st);
        }
        if (rc) return rc;
        SGPR_HIP(hipMemcpyAsync(w.data(), dw.p, n * sizeof(double), hipMemcpyDeviceToHost, f->st));
        SGPR_HIP(hipStreamSynchronize(f->st));
        const double s2 = std::fabs(f->sig2n);
        for (size_t i = 0; i < n; ++i) w[i] += s2 * v[i];
        double q;
        normalise(w, v, q);
        ch_max = lmax > 0.0 ? std::fabs(q - lmax) / q : 1.0;
        lmax = q;
    }
    v = v0;
    for (int it = 0; it < iters; ++it) {                // ---- 1 / lambda_min
        SGPR_HIP(hipMemcpyAsync(dw.p, v.data(), n * sizeof(double), hipMemcpyHostToDevice, f->st));
        if ((rc = potrs_vec(f->n, f->dA, n, f->work, dw.as<double>(), f->st))) return rc;
        if ((rc = solve_status(f->n, f->dA, n, f->work, f->st))) return rc;
        SGPR_HIP(hipMemcpyAsync(w.data(), dw.p, n * sizeof(double), hipMemcpyDeviceToHost, f->st));
        SGPR_HIP(hipStreamSynchronize(f->st));
        double q;
        normalise(w, v, q);
        ch_min = lmin_inv > 0.0 ? std::fabs(q - lmin_inv) / q : 1.0;
        lmin_inv = q;
    }
    (void)hyp1; (void)nh1;
    out4[0] = lmax;
    out4[1] = lmin_inv > 0.0 ? 1.0 / lmin_inv : 0.0;
    out4[2] = lmax * lmin_inv;
    out4[3] = ch_max > ch_min ? ch_max : ch_min;
    return 0;
}

int sgpr_fit_trim(sgpr_fit_t f)
{
    if (!f) { set_error("null fit"); return SGPR_E_ARG; }
    if (f->rhs_scratch) {
        SGPR_HIP(hipStreamSynchronize(f->st));
        (void)hipFree(f->rhs_scratch);
        f->rhs_scratch = nullptr;
        f->rhs_scratch_bytes = 0;
    }
    return 0;
}

int sgpr_fit_stage_ms(sgpr_fit_t f, double *build_ms, double *factor_ms, double *solve_ms)
{
    if (!f) { set_error("null fit"); return SGPR_E_ARG; }
    SGPR_HIP(hipStreamSynchronize(f->st));
    double *outs[3] = {build_ms, factor_ms, solve_ms};
    for (int s = 0; s < 3; ++s) {
        if (!outs[s]) continue;
        float ms = -1.0f;
        if (f->timed[s]) SGPR_HIP(hipEventElapsedTime(&ms, f->ev[2 * s], f->ev[2 * s + 1]));
        *outs[s] = ms;
    }
    return 0;
}

int sgpr_fit_solve_rhs_ms(sgpr_fit_t f, double *ms_out)
{
    if (!f || !ms_out) { set_error("null argument"); return SGPR_E_ARG; }
    SGPR_HIP(hipStreamSynchronize(f->st));
    float ms = -1.0f;
    if (f->timed[3]) SGPR_HIP(hipEventElapsedTime(&ms, f->ev[6], f->ev[7]));
    *ms_out = ms;
    return 0;
}

int sgpr_fit_device_ptrs(sgpr_fit_t f, void **dA, size_t *lda, void **dalpha)
{
    if (!f) { set_error("null fit"); return SGPR_E_ARG; }
    if (dA) *dA = f->dA;
    if (lda) *lda = (size_t)f->n;
    if (dalpha) *dalpha = f->dalpha;
    return 0;
}

/* ---- many small fits in one launch ------------------------------------------------------- */

int sgpr_fit_batch_max_order(void) { return fit_batch_max_order(); }

int sgpr_fit_batch(int family, int nbatch, int n_pts, const double *x, const double *y, const double *z,
                   const double *hyp, int nhyp, const double *sig2n, unsigned flags, double *alpha, double *nll,
                   int *info)
{
    int rc = need_device();
    if (rc) return rc;
    return fit_batch(family, nbatch, n_pts, x, y, z, hyp, nhyp, sig2n, flags, alpha, nll, info);
}

/* ---- device-pointer primitives ---------------------------------------------------------- */

int sgpr_gram_pairs_dev(int family, int mi, int mj, const double *xb, const double *yb,
                        const double *xa, const double *ya, const double *hyp, int nhyp, double *qq,
                        double *Pq, double *qP, double *PP, size_t ld, long diag_off, double noise,
                        unsigned flags, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    KConst kc;
    if ((rc = make_kconst(family, hyp, nhyp, &kc))) return rc;
    return gram_pairs(family, mi, mj, xb, yb, xa, ya, kc, qq, Pq, qP, PP, ld, diag_off, std::fabs(noise),
                      flags, static_cast<hipStream_t>(stream));
}

int sgpr_gram_reg_dev(int family, int mi, int mj, const double *xb, const double *yb, const double *xa,
                      const double *ya, const double *hyp, int nhyp, double *G, size_t ld,
                      long diag_off, double noise, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    KConst kc;
    if ((rc = make_kconst(family, hyp, nhyp, &kc))) return rc;
    return gram_reg(family, mi, mj, xb, yb, xa, ya, kc, G, ld, diag_off, std::fabs(noise),
                    static_cast<hipStream_t>(stream));
}

int sgpr_gram_nd_dev(int family, int d, int mi, int mj, const double *Xb, size_t ldxb, const double *Xa,
                     size_t ldxa, const double *hyp, int nhyp, double *K, size_t ld, size_t rstride,
                     size_t cstride, long diag_off, double noise, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return gram_nd(family, d, mi, mj, Xb, ldxb, Xa, ldxa, hyp, nhyp, K, ld, rstride, cstride, diag_off,
                   std::fabs(noise), static_cast<hipStream_t>(stream));
}

int sgpr_gram_nd_sel_dev(int family, int d, int mi, int mj, const double *Xb, size_t ldxb, const double *Xa,
                         size_t ldxa, const double *hyp, int nhyp, double *K, size_t ld, const long *roff,
                         const long *coff, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return gram_nd_sel(family, d, mi, mj, Xb, ldxb, Xa, ldxa, hyp, nhyp, K, ld, roff, coff, static_cast<hipStream_t>(stream));
}

size_t sgpr_potrf_workspace(int n) { return potrf_workspace(n); }
size_t sgpr_potrf_inverses_bytes(int n) { return n <= 0 ? 0 : (size_t)((n + LEAF - 1) / LEAF) * LEAF * LEAF * sizeof(double); }

int sgpr_family_has_p(int family) { return family_has_p(family) ? 1 : 0; }

int sgpr_release_device_streams(int device)
{
    return release_device_streams(device);      // (touches the device only if this library has streams on it)
}

int sgpr_potrf_dev(int n, double *A, size_t lda, void *work, size_t lwork, int *dinfo, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return potrf(n, A, lda, work, lwork, dinfo, static_cast<hipStream_t>(stream));
}

int sgpr_potrf_info_dev(int info, void *stream)
{
    if (info >= 0) return info;
    if (info == POTRF_HANDOFF_TIMEOUT) (void)potrf_queue_mark_failed(static_cast<hipStream_t>(stream));
    return info_status(info);
}

int sgpr_trsm_rlt_dev(int m, int n, const double *L, size_t ldl, double *B, size_t ldb,
                      const void *work, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return trsm_rlt(m, n, L, ldl, B, ldb, work, static_cast<hipStream_t>(stream));
}

int sgpr_gemm_nt_dev(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
                     size_t ldb, double beta, double *C, size_t ldc, int lower, long diag_off,
                     void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return gemm_nt(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lower, diag_off,
                   static_cast<hipStream_t>(stream));
}

int sgpr_gemm_nt_bc_dev(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
                        size_t ldb, double beta, double *C, size_t ldc, int blk, int pr, int pi, int pc,
                        int pj, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    const int bc[5] = {blk, pr, pi, pc, pj};
    return gemm_nt_bc(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 1, bc, static_cast<hipStream_t>(stream));
}

int sgpr_trsv_dev(int n, const double *L, size_t ldl, void *work, double *b, int trans, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return trsv(n, L, ldl, work, b, trans, static_cast<hipStream_t>(stream));
}

int sgpr_gemv_sub_dev(int trans, int m, int k, const double *A, size_t lda, const double *x, double *y,
                      void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    if (m < 0 || k < 0 || (m > 0 && lda < (size_t)m)) { set_error("gemv: bad shape"); return SGPR_E_ARG; }
    return trans ? gemv_t_sub(m, k, A, lda, x, y, static_cast<hipStream_t>(stream))
                 : gemv_n_sub(m, k, A, lda, x, y, static_cast<hipStream_t>(stream));
}

int sgpr_gemm_nn_dev(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B, size_t ldb,
                     double beta, double *C, size_t ldc, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return gemm_nn(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, static_cast<hipStream_t>(stream));
}

int sgpr_trsm_rl_dev(int m, int n, const double *L, size_t ldl, double *B, size_t ldb, const void *work, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return trsm_rl(m, n, L, ldl, B, ldb, work, static_cast<hipStream_t>(stream));
}

namespace sgpr { namespace {
// cnt blocks of rows x cols doubles, block i from src + i * sstep (leading dimension lds) to dst + i * dstep (ldd): the panel
// packing / regrouping copies of the block-cyclic driver in one launch (rows fastest: 512-byte runs per wave)
__global__ __launch_bounds__(256) void copy_blocks_kernel(int rows, int cols, int cnt, const double *src, size_t lds, size_t sstep,
                                                          double *dst, size_t ldd, size_t dstep)
{
    const int i = blockIdx.z;
    const double *s = src + (size_t)i * sstep;
    double *d = dst + (size_t)i * dstep;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    for (int c = blockIdx.y; c < cols; c += gridDim.y) d[(size_t)r + (size_t)c * ldd] = s[(size_t)r + (size_t)c * lds];
}
} }

int sgpr_copy_blocks_dev(int rows, int cols, int cnt, const double *src, size_t lds, size_t sstep, double *dst, size_t ldd,
                         size_t dstep, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    if (rows < 0 || cols < 0 || cnt < 0 || (rows > 0 && (lds < (size_t)rows || ldd < (size_t)rows))) { set_error("copy_blocks: bad shape"); return SGPR_E_ARG; }
    if (rows == 0 || cols == 0 || cnt == 0) return 0;
    if (cnt > 65535) { set_error("copy_blocks: more than 65535 blocks"); return SGPR_E_ARG; }
    const dim3 grid((unsigned)((rows + 255) / 256), (unsigned)std::min(cols, 1024), (unsigned)cnt);
    hipLaunchKernelGGL(sgpr::copy_blocks_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), rows, cols, cnt, src, lds, sstep,
                       dst, ldd, dstep);
    SGPR_CHECK_LAUNCH();
    return 0;
}

int sgpr_predict_rows_dev(int family, int m, const double *q, const double *P, int n0, const double *xtrain,
                          const double *ytrain, const double *hyp, int nhyp, const double *alpha,
                          double *out_p, double *out_q, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    KConst kc;
    if ((rc = make_kconst(family, hyp, nhyp, &kc))) return rc;
    return predict_rows(family, m, q, P, n0, xtrain, ytrain, kc, alpha, out_p, out_q,
                        static_cast<hipStream_t>(stream));
}

int sgpr_predict_nd_dev(int family, int d, int m, const double *Xt, size_t ldxt, int n0, const double *Xtrain,
                        size_t ldxtr, const double *hyp, int nhyp, const double *alpha, double *out, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return predict_nd(family, d, m, Xt, ldxt, n0, Xtrain, ldxtr, hyp, nhyp, alpha, out, static_cast<hipStream_t>(stream));
}

int sgpr_predict_reg_dev(int family, int m, const double *q, const double *P, int n0, const double *xtrain,
                         const double *ytrain, const double *hyp, int nhyp, const double *alpha,
                         double *out, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    KConst kc;
    if ((rc = make_kconst(family, hyp, nhyp, &kc))) return rc;
    return predict_reg(family, m, q, P, n0, xtrain, ytrain, kc, alpha, out, static_cast<hipStream_t>(stream));
}

/* applymap / applymap_henon (functions/func.py:216-260) and the per-example variants for all Ntest
 * orbits, every time step on the device.  alpha = Kyinv ztrain (2 n0), alphap = Kyinvp ztrainp (n0p);
 * qmap, pmap, pdiff: [nm][ntest] C-ordered host arrays (row 0 = initial conditions), pdiff optional.
 * mode: SGPR_MAP_* bits. */
int sgpr_applymap_host(int family, int mode, int nm, int ntest, const double *hyp, int nhyp, int n0,
                       const double *xtrain, const double *ytrain, const double *alpha, const double *hypp,
                       int nhypp, int n0p, const double *xtrainp, const double *ytrainp, const double *alphap,
                       const double *Q0, const double *P0, double *qmap, double *pmap, double *pdiff)
{
    int rc = need_device();
    if (rc) return rc;
    const bool expl = (mode & SGPR_MAP_EXPLICIT) != 0;
    if (expl) n0p = 0;                       /* no first-guess GP in the explicit map */
    if (nm < 1 || ntest < 0 || n0 < 0 || n0p < 0 || (mode & ~15) || (expl && (mode & SGPR_MAP_LOSS_NEGP)) || !qmap || !pmap) {
        set_error("applymap: bad arguments");
        return SGPR_E_ARG;
    }
    KConst kc, kcp{};
    if ((rc = make_kconst(family, hyp, nhyp, &kc))) return rc;
    if (!expl && (rc = make_kconst(family, hypp, nhypp, &kcp))) return rc;
    if (ntest == 0) return 0;
    if (!Q0 || !P0 || (n0 > 0 && (!xtrain || !ytrain || !alpha)) || (n0p > 0 && (!xtrainp || !ytrainp || !alphap))) {
        set_error("applymap: null argument");
        return SGPR_E_ARG;
    }
    DevBuf x, y, al, xp, yp, alp, q0, p0, qm, pm, pd, tw;
    hipStream_t st = nullptr;
    const size_t out_bytes = (size_t)nm * ntest * sizeof(double);
    if ((rc = tw.alloc(applymap_team_ws(ntest, n0)))) return rc;
    if ((rc = upload(x, xtrain, n0, st)) || (rc = upload(y, ytrain, n0, st)) || (rc = upload(al, alpha, 2 * (size_t)n0, st)) ||
        (rc = upload(xp, xtrainp, n0p, st)) || (rc = upload(yp, ytrainp, n0p, st)) || (rc = upload(alp, alphap, n0p, st)) ||
        (rc = upload(q0, Q0, ntest, st)) || (rc = upload(p0, P0, ntest, st)) || (rc = qm.alloc(out_bytes)) ||
        (rc = pm.alloc(out_bytes)) || (pdiff && (rc = pd.alloc(out_bytes))))
        return rc;
    rc = applymap(family, mode, nm, ntest, n0, x.as<double>(), y.as<double>(), kc, al.as<double>(), n0p,
                  xp.as<double>(), yp.as<double>(), kcp, alp.as<double>(), q0.as<double>(), p0.as<double>(),
                  qm.as<double>(), pm.as<double>(), pdiff ? pd.as<double>() : nullptr, tw.p, st);
    if (rc) return rc;
    SGPR_HIP(hipMemcpyAsync(qmap, qm.p, out_bytes, hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipMemcpyAsync(pmap, pm.p, out_bytes, hipMemcpyDeviceToHost, st));
    if (pdiff) SGPR_HIP(hipMemcpyAsync(pdiff, pd.p, out_bytes, hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    return applymap_status(tw.p, ntest, n0);
}

int sgpr_trim(void)
{
    (void)fit_batch_trim();
    return potrf_trim();
}

int sgpr_profile_begin(void) { return gemm_profile_begin(); }
int sgpr_profile_end(double *out12)
{
    if (!out12) { set_error("null argument"); return SGPR_E_ARG; }
    return gemm_profile_end(out12);
}

int sgpr_profile_launches(double *buf, int max_records) { return gemm_profile_launches(buf, max_records); }

int sgpr_potrs_vec_dev(int n, const double *L, size_t ldl, void *work, double *b, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return potrs_vec(n, L, ldl, work, b, static_cast<hipStream_t>(stream));
}

int sgpr_solve_status_dev(int n, const double *L, size_t ldl, const void *work, void *stream)
{
    int rc = need_device();
    if (rc) return rc;
    return solve_status(n, L, ldl, work, static_cast<hipStream_t>(stream));
}

}  // extern "C"

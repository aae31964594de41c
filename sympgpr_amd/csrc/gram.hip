// gram.hip -- Gram-matrix build for gfx950 (MI355X).
//
// Replaces the pair loop of the reference's build_K / buildKreg
// (python/05_tokamak/SympGPR/sympgpr.f90:25-37, :54-59; pure-Python form
// python/01_pendulum/implicit/func.py:55-64) and the scalar kernels they call
// (kernels.f90:1-11,58-94 and the kernels_sq / kernels_sum / period-unknown variants).
//
// Roofline: HBM write.  One pair (i,j) yields four fp64 entries = 32 B written and costs one
// exp + one sincos + ~20 FMA, all on the fp64 VALU.  Layout decisions:
//   * column-major output, so the row index i is the contiguous one: a thread owns TWO
//     consecutive rows (one 16-B store per part), a wave writes 1 KiB contiguous per store
//     instruction, a workgroup owns a 512-row x 16-column pair tile = 4 x 64 KiB of K;
//   * the 16 column points of the tile are staged once in LDS and read back as broadcasts;
//   * sig scaling, the |sig2n| diagonal and the lower-triangle cut are fused here (the
//     reference spends three more n^2 passes on them: sympgpr.f90:37, func.py:192).
#include <atomic>
#include "common.h"
#include "devmath.h"
#include "pair_eval.h"

namespace sgpr {

namespace {

constexpr int GT = 256;        // threads per workgroup
constexpr int TI = 2 * GT;     // pair rows per tile (2 per thread)
constexpr int TJ = 16;         // pair columns per tile (32 -> 16: n = 16384 0.49 -> 0.46 ms, no change at n = 131072)

typedef double double2_t __attribute__((ext_vector_type(2)));

struct GramArgs {
    int mi, mj;
    const double *xb, *yb, *xa, *ya;
    double *dst[4];  // qq, Pq, qP, PP
    size_t ld;
    long diag_off;
    double noise;
    unsigned flags;
    KConst kc;
};

using namespace pairf;

template <int FAM, bool OCML, int DL>
__global__ __launch_bounds__(GT) void gram_pairs_kernel(const GramArgs a)
{
    __shared__ double sxa[TJ], sya[TJ];
    const int i0 = blockIdx.x * TI;
    const int j0 = blockIdx.y * TJ;
    const int t = threadIdx.x;
    const int nj = min(TJ, a.mj - j0);
    if (t < nj) {
        sxa[t] = a.xa[j0 + t];
        sya[t] = a.ya[j0 + t];
    }
    // which parts does this tile write?  (block-uniform)
    const bool lower = a.flags & SGPR_G_LOWER;
    const long last_row = (long)min(i0 + TI, a.mi) - 1 + a.diag_off;
    const bool on_or_below = !lower || last_row >= (long)j0;
    double *const pqq = (a.flags & SGPR_G_QQ) && on_or_below ? a.dst[0] : nullptr;
    double *const pPq = (a.flags & SGPR_G_PQ) ? a.dst[1] : nullptr;
    double *const pqP = (a.flags & SGPR_G_QP) && !lower ? a.dst[2] : nullptr;
    double *const pPP = (a.flags & SGPR_G_PP) && on_or_below ? a.dst[3] : nullptr;
    __syncthreads();
    if (!pqq && !pPq && !pqP && !pPP) return;

    const int i = i0 + 2 * t;
    const bool v0 = i < a.mi, v1 = i + 1 < a.mi;
    const double xb0 = v0 ? a.xb[i] : 0.0, yb0 = v0 ? a.yb[i] : 0.0;
    const double xb1 = v1 ? a.xb[i + 1] : 0.0, yb1 = v1 ? a.yb[i + 1] : 0.0;
    // 16-B vector stores need every part's (i, j) address 16-B aligned: i is even, so the
    // base pointers and ld decide (block-uniform); the ragged last row tile goes scalar.
    const bool vec = (i0 + TI <= a.mi) && ((a.ld & 1) == 0) &&
                     ((((uintptr_t)a.dst[0] | (uintptr_t)a.dst[1] | (uintptr_t)a.dst[2] |
                        (uintptr_t)a.dst[3]) & 15) == 0);
    const long d0 = (long)i + a.diag_off;  // global column that is "diagonal" for row i
    const double noise = a.noise;

    if (vec) {
#pragma unroll 2
        for (int jj = 0; jj < nj; ++jj) {
            const double xa = sxa[jj], ya = sya[jj];
            double kxx0, kxy0, kyy0, kxx1, kxy1, kyy1;
            pair_any<FAM, OCML, DL>(xa, ya, xb0, yb0, a.kc, kxx0, kxy0, kyy0);
            pair_any<FAM, OCML, DL>(xa, ya, xb1, yb1, a.kc, kxx1, kxy1, kyy1);
            const long j = j0 + jj;
            const double n0 = (d0 == j) ? noise : 0.0, n1 = (d0 + 1 == j) ? noise : 0.0;
            const size_t off = (size_t)i + (size_t)j * a.ld;
            if (pqq) *reinterpret_cast<double2_t *>(pqq + off) = double2_t{kxx0 + n0, kxx1 + n1};
            if (pPq) *reinterpret_cast<double2_t *>(pPq + off) = double2_t{kxy0, kxy1};
            if (pqP) *reinterpret_cast<double2_t *>(pqP + off) = double2_t{kxy0, kxy1};
            if (pPP) *reinterpret_cast<double2_t *>(pPP + off) = double2_t{kyy0 + n0, kyy1 + n1};
        }
    } else {
        for (int jj = 0; jj < nj; ++jj) {
            const double xa = sxa[jj], ya = sya[jj];
            double kxx0, kxy0, kyy0, kxx1, kxy1, kyy1;
            pair_any<FAM, OCML, DL>(xa, ya, xb0, yb0, a.kc, kxx0, kxy0, kyy0);
            pair_any<FAM, OCML, DL>(xa, ya, xb1, yb1, a.kc, kxx1, kxy1, kyy1);
            const long j = j0 + jj;
            const double n0 = (d0 == j) ? noise : 0.0, n1 = (d0 + 1 == j) ? noise : 0.0;
            const size_t off = (size_t)i + (size_t)j * a.ld;
            if (v0) {
                if (pqq) pqq[off] = kxx0 + n0;
                if (pPq) pPq[off] = kxy0;
                if (pqP) pqP[off] = kxy0;
                if (pPP) pPP[off] = kyy0 + n0;
            }
            if (v1) {
                if (pqq) pqq[off + 1] = kxx1 + n1;
                if (pPq) pPq[off + 1] = kxy1;
                if (pqP) pqP[off + 1] = kxy1;
                if (pPP) pPP[off + 1] = kyy1 + n1;
            }
        }
    }
}

struct RegArgs {
    int mi, mj;
    const double *xb, *yb, *xa, *ya;
    double *G;
    size_t ld;
    long diag_off;
    double noise;
    KConst kc;
};

template <int FAM, bool OCML, int DL>
__global__ __launch_bounds__(GT) void gram_reg_kernel(const RegArgs a)
{
    __shared__ double sxa[TJ], sya[TJ];
    const int i0 = blockIdx.x * TI;
    const int j0 = blockIdx.y * TJ;
    const int t = threadIdx.x;
    const int nj = min(TJ, a.mj - j0);
    if (t < nj) {
        sxa[t] = a.xa[j0 + t];
        sya[t] = a.ya[j0 + t];
    }
    __syncthreads();
    const int i = i0 + 2 * t;
    const bool v0 = i < a.mi, v1 = i + 1 < a.mi;
    const double xb0 = v0 ? a.xb[i] : 0.0, yb0 = v0 ? a.yb[i] : 0.0;
    const double xb1 = v1 ? a.xb[i + 1] : 0.0, yb1 = v1 ? a.yb[i + 1] : 0.0;
    const bool vec = (i0 + TI <= a.mi) && ((a.ld & 1) == 0) && (((uintptr_t)a.G & 15) == 0);
    const long d0 = (long)i + a.diag_off;
    for (int jj = 0; jj < nj; ++jj) {
        const double xa = sxa[jj], ya = sya[jj];
        const long j = j0 + jj;
        const double k0 = a.kc.sig * kern_any<FAM, OCML, DL>(xa, ya, xb0, yb0, a.kc) +
                          ((d0 == j) ? a.noise : 0.0);
        const double k1 = a.kc.sig * kern_any<FAM, OCML, DL>(xa, ya, xb1, yb1, a.kc) +
                          ((d0 + 1 == j) ? a.noise : 0.0);
        const size_t off = (size_t)i + (size_t)j * a.ld;
        if (vec) {
            *reinterpret_cast<double2_t *>(a.G + off) = double2_t{k0, k1};
        } else {
            if (v0) a.G[off] = k0;
            if (v1) a.G[off + 1] = k1;
        }
    }
}

// kernels.<name>_num, elementwise (sig = 1).  DL selects the length-scale derivative family
// (dkdlx_num, d3kdxdx0dlx_num, ... kernels.f90:133-231).
template <int FAM, int DL>
__global__ void kernel_eval_kernel(int which, int m, const double *xa, const double *ya,
                                   const double *xb, const double *yb, const KConst kc, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double r;
    if (which == SGPR_K_KERN) {
        r = kern_any<FAM, false, DL>(xa[i], ya[i], xb[i], yb[i], kc);
    } else {
        double kxx, kxy, kyy;
        pair_any<FAM, false, DL>(xa[i], ya[i], xb[i], yb[i], kc, kxx, kxy, kyy);
        r = which == SGPR_K_DXDX0 ? kxx : (which == SGPR_K_DYDY0 ? kyy : kxy);
    }
    out[i] = r;
}

// The seven generated functions no caller of the reference uses (first derivatives, third derivatives
// with respect to y_b: kernels.f90:12-57,95-132 and the sibling files), for the completeness of the
// `kernels` module.  Not a hot path: plain device-libs sin / cos / exp.
template <int FAM>
__device__ double extra_eval(int which, double xa, double ya, double xb, double yb, const KConst &kc)
{
    if constexpr (FAM == SGPR_FAM_USER) return gen::extra<SGPR_FAM_USER>(which, xa, ya, xb, yb, kc.lx, kc.ly, kc.p);
    const double lx2 = kc.lx2, ly2 = kc.ly2, dy = ya - yb;
    if constexpr (FAM == SGPR_FAM_A || FAM == SGPR_FAM_D) {
        const double h = FAM == SGPR_FAM_A ? 0.5 * xa - 0.5 * xb : kc.p * (xa - xb);
        const double hs = FAM == SGPR_FAM_A ? 0.5 : kc.p;
        const double s = sin(h), c = cos(h), cd = cos(2.0 * h);
        const double E = exp(-0.5 * (lx2 * dy * dy + ly2 * s * s) / (lx2 * ly2));
        switch (which) {
        case SGPR_K_DX: return -hs * E * s * c / lx2;
        case SGPR_K_DY: return -dy * E / ly2;
        case SGPR_K_DX0: return hs * E * s * c / lx2;
        case SGPR_K_DY0: return dy * E / ly2;
        case SGPR_K_DXDX0DY0: return hs * hs * dy * (lx2 * cd - s * s * c * c) * E / (lx2 * lx2 * ly2);
        case SGPR_K_DYDY0DY0: return (3.0 * ly2 - dy * dy) * dy * E / (ly2 * ly2 * ly2);
        default: return hs * (ly2 - dy * dy) * E * s * c / (lx2 * ly2 * ly2);   // SGPR_K_DXDY0DY0
        }
    } else if constexpr (FAM == SGPR_FAM_B) {
        const double h = 0.5 * xa - 0.5 * xb, s = sin(h), c = cos(h);
        const double ex = exp(-0.5 * s * s / lx2);
        const double ey = exp((-0.5 * ya * ya + ya * yb - 0.5 * yb * yb) / ly2);
        switch (which) {
        case SGPR_K_DX: return -0.5 * ex * s * c / lx2;
        case SGPR_K_DY: return -dy * ey / ly2;
        case SGPR_K_DX0: return 0.5 * ex * s * c / lx2;
        case SGPR_K_DY0: return dy * ey / ly2;
        case SGPR_K_DYDY0DY0: return (3.0 * ly2 - dy * dy) * dy * ey / (ly2 * ly2 * ly2);
        default: return 0.0;                                                    // the mixed ones vanish
        }
    } else {
        const double dx = xa - xb;
        const double E = exp(-0.5 * (lx2 * dy * dy + ly2 * dx * dx) / (lx2 * ly2));
        switch (which) {
        case SGPR_K_DX: return -dx * E / lx2;
        case SGPR_K_DY: return -dy * E / ly2;
        case SGPR_K_DX0: return dx * E / lx2;
        case SGPR_K_DY0: return dy * E / ly2;
        case SGPR_K_DXDX0DY0: return (lx2 - dx * dx) * dy * E / (lx2 * lx2 * ly2);
        case SGPR_K_DYDY0DY0: return (3.0 * ly2 - dy * dy) * dy * E / (ly2 * ly2 * ly2);
        default: return (ly2 - dy * dy) * dx * E / (lx2 * ly2 * ly2);
        }
    }
}

template <int FAM>
__global__ void kernel_extra_kernel(int which, int m, const double *xa, const double *ya, const double *xb,
                                    const double *yb, const KConst kc, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = extra_eval<FAM>(which, xa[i], ya[i], xb[i], yb[i], kc);
}

// K*(2 x 2n0) . alpha for one test point per workgroup (sympgpr.f90:75-86, :112-124 with
// alpha = Kyinv ztrain cached): row 1 -> out_p, row 2 -> out_q.
template <int FAM>
__global__ __launch_bounds__(GT) void predict_rows_kernel(int n0, const double *q, const double *P,
                                                          const double *xtr, const double *ytr,
                                                          const KConst kc, const double *alpha,
                                                          double *out_p, double *out_q)
{
    const int k = blockIdx.x;
    const double xb = q[k], yb = P[k];
    double r1 = 0.0, r2 = 0.0;
    for (int j = threadIdx.x; j < n0; j += GT) {
        double kxx, kxy, kyy;
        pair_eval<FAM, false>(xtr[j], ytr[j], xb, yb, kc, kxx, kxy, kyy);
        const double a1 = alpha[j], a2 = alpha[n0 + j];
        r1 += kxx * a1 + kxy * a2;
        r2 += kxy * a1 + kyy * a2;
    }
    __shared__ double s1[GT / 64], s2[GT / 64];
    for (int o = 32; o > 0; o >>= 1) {
        r1 += __shfl_down(r1, o, 64);
        r2 += __shfl_down(r2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s1[threadIdx.x >> 6] = r1;
        s2[threadIdx.x >> 6] = r2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < GT / 64; ++w) {
            t1 += s1[w];
            t2 += s2[w];
        }
        out_p[k] = t1;
        out_q[k] = t2;
    }
}

// Kstar(1 x n0) . alpha_p with the scalar kernel (sympgpr.f90:62-73 guessP)
template <int FAM>
__global__ __launch_bounds__(GT) void predict_reg_kernel(int n0, const double *q, const double *P,
                                                         const double *xtr, const double *ytr,
                                                         const KConst kc, const double *alpha,
                                                         double *out)
{
    const int k = blockIdx.x;
    const double xb = q[k], yb = P[k];
    double r = 0.0;
    for (int j = threadIdx.x; j < n0; j += GT)
        r += kc.sig * kern_eval<FAM, false>(xtr[j], ytr[j], xb, yb, kc) * alpha[j];
    __shared__ double s1[GT / 64];
    for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o, 64);
    if ((threadIdx.x & 63) == 0) s1[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < GT / 64; ++w) t += s1[w];
        out[k] = t;
    }
}

// ---- applymap: the whole symplectic-map iteration of one orbit inside one workgroup ------------
// functions/func.py:216-260 (applymap / applymap_henon) with calcP / calcQ / guessP of
// sympgpr.f90:62-125 inlined: per time step, P_new is the root of f(P) = pGP(q, P) - p + P started
// from the regular-GP guess (the reference runs MINPACK hybrd1, tol 1e-13, per point and step, each
// residual an O(n^2) matmul with Kyinv); here alpha = Kyinv ztrain is cached, a residual is one
// block-wide reduction over the training points, and all nm steps run without leaving the GPU.
constexpr int MAP_TEAM_T = 256;     // threads of a team member
constexpr int MAP_TEAM_MAX = 16;    // members of a team at most
constexpr int MAP_STAGE = 5;        // rounds of TT training points a member keeps in LDS (7 doubles per point: 70 KB at 256 threads)
struct MapArgs {
    int nm, ntest, n0, n0p, mode, maxiter;
    int S;                                // workgroups per orbit (the team): each sums its share of the training points
    unsigned long long *tw;               // team exchange words: [orbit][member][parity][2] 16-byte granules {value, sequence}, zero on entry
    int *err;                             // set when a team member gave up waiting for another
    double tol;
    const double *xtr, *ytr, *alpha;      // symplectic GP: n0 points, alpha 2 n0
    const double *xtrp, *ytrp, *alphap;   // regular GP (guess): n0p points
    const double *Q0, *P0;
    double *qmap, *pmap, *pdiff;          // [nm][ntest], C order (numpy zeros([nm, Ntest])); pdiff may be null
    KConst kc, kcp;
};

// `sh`: 2 * (TT / 64) doubles that the call before did NOT use (the callers alternate between two: a wave that is still reading
// the sums of call n cannot be overtaken by the writes of call n + 2, because call n + 1's barrier lies between)
template <int TT>
__device__ __forceinline__ void block_sum2(double &a, double &b, double *sh)
{
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[2 * (threadIdx.x >> 6)] = a;
        sh[2 * (threadIdx.x >> 6) + 1] = b;
    }
    __syncthreads();
    double x = 0.0, y = 0.0;
#pragma unroll
    for (int w = 0; w < TT / 64; ++w) { x += sh[2 * w]; y += sh[2 * w + 1]; }
    a = x; b = y;
}

// 16-byte {value, sequence number} granules, written by one write-through store and read by one load that passes the vector L1
// (MI355X_MICROARCH.md, inter-workgroup visibility: 16-byte sc1 halves were observed untorn): the value and the word that says
// which residual it belongs to arrive together, so a member that finds the expected sequence number has the value.
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void granule_store(unsigned long long *p, double v, unsigned seq)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint4_t w = {(unsigned)b, (unsigned)(b >> 32), seq, 0u};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
}
// both granules of a pair in flight together: one memory round trip per look
__device__ __forceinline__ void granule_load2(const unsigned long long *p, uint4_t &u, uint4_t &v)
{
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(u), "=&v"(v) : "v"(p) : "memory");
}

// One workgroup per (orbit, team member).  With a.S == 1 this is the round-2 kernel: one workgroup runs the whole iteration
// of its orbit.  For large training sets (the drivers use 20 - 80 points; BASELINE config 05 has 16384) S workgroups share an
// orbit: every residual is summed in S parts, exchanged through a.tw and added up in member order by every member alike -- all
// members then hold the same bits, take the same branches and need no leader.  Ntest = 37 orbits no longer mean 37 CUs.
// TT threads: 256 for one workgroup per orbit (the drivers' sizes: latency of a few dozen points), MAP_TEAM_T for the teams
template <int FAM, int TT>
__global__ __launch_bounds__(TT) void applymap_kernel(const MapArgs a)
{
    __shared__ double sh[2][2 * (TT / 64)];
    __shared__ double sp[2][2][MAP_TEAM_MAX];           // [parity of the call]
    const int S = a.S, k = blockIdx.x / S, me = blockIdx.x - k * S;
    unsigned seq = 0;
    bool lost = false;                                  // a team member did not answer in time: the orbit is lost (NaN), a.err says why
    // (x, y) := sum over the team of every thread's (x, y), identical bits in every member: the workgroup's own sum first, then
    // ONE pair of granules per member, collected by lanes 0 .. S - 1.  (Tried: every wave publishing its own part, 4 S and 16 S
    // granule pairs to collect -- no gain at 256 threads, 40 instead of 65 G pair evaluations per second at 1024.)
    __shared__ int sh_lost;
    if (threadIdx.x == 0) sh_lost = 0;
    __syncthreads();
    auto team_sum2 = [&](double &x, double &y) {
        ++seq;
        block_sum2<TT>(x, y, sh[seq & 1u]);
        if (S == 1) return;
        if (lost) {                                     // sticky: one timeout ends the orbit, nobody waits 2 s per sum after it
            x = y = __builtin_nan("");
            return;
        }
        if (threadIdx.x == 0) {
            unsigned long long *mine = a.tw + (((size_t)k * S + me) * 2 + (seq & 1u)) * 4;
            granule_store(mine, x, seq);
            granule_store(mine + 2, y, seq);
        }
        if (threadIdx.x < (unsigned)S) {
            const unsigned long long *theirs = a.tw + (((size_t)k * S + threadIdx.x) * 2 + (seq & 1u)) * 4;
            uint4_t u, v;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned it = 0;
            bool ok = true;
            for (;;) {
                granule_load2(theirs, u, v);
                if (u[2] == seq && v[2] == seq) break;
                if ((++it & 255u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { ok = false; break; }    // 2 s
            }
            sp[seq & 1u][0][threadIdx.x] = ok ? __longlong_as_double((long long)(((unsigned long long)u[1] << 32) | u[0])) : __builtin_nan("");
            sp[seq & 1u][1][threadIdx.x] = ok ? __longlong_as_double((long long)(((unsigned long long)v[1] << 32) | v[0])) : __builtin_nan("");
            if (!ok) { atomicExch(a.err, 1); sh_lost = 1; }
        }
        __syncthreads();
        lost = sh_lost != 0;                            // (block-uniform; sh_lost is only ever raised)
        double sx = 0.0, sy = 0.0;
        for (int m = 0; m < S; ++m) { sx += sp[seq & 1u][0][m]; sy += sp[seq & 1u][1][m]; }
        x = sx; y = sy;                                 // (no barrier behind the reads: the next call writes the other halves)
    };
    // This member's training points never change: they are staged in LDS once (a residual is a handful of points per thread, and
    // their loads -- four per point, L2 latency each round -- were two thirds of its time at N0 = 16 384: 1.2 us per point and
    // thread against ~0.4 of arithmetic).  Same points, same order per thread: same bits.  Slices that do not fit stay in memory.
    __shared__ double st[7][MAP_STAGE * TT];            // x, y, alpha (two halves) of the symplectic GP; x, y, alpha of the guess
    auto rounds_of = [&](int n) { return n > me * TT ? (n - me * TT + S * TT - 1) / (S * TT) : 0; };
    const int nr = rounds_of(a.n0), nrp = rounds_of(a.n0p);
    const bool staged = nr <= MAP_STAGE && nrp <= MAP_STAGE;
    if (staged) {
        for (int i = 0; i < nr; ++i) {
            const int j = me * TT + (int)threadIdx.x + i * S * TT;
            if (j < a.n0) {
                st[0][i * TT + threadIdx.x] = a.xtr[j]; st[1][i * TT + threadIdx.x] = a.ytr[j];
                st[2][i * TT + threadIdx.x] = a.alpha[j]; st[3][i * TT + threadIdx.x] = a.alpha[a.n0 + j];
            }
        }
        for (int i = 0; i < nrp; ++i) {
            const int j = me * TT + (int)threadIdx.x + i * S * TT;
            if (j < a.n0p) {
                st[4][i * TT + threadIdx.x] = a.xtrp[j]; st[5][i * TT + threadIdx.x] = a.ytrp[j]; st[6][i * TT + threadIdx.x] = a.alphap[j];
            }
        }
        // (every thread reads back what it wrote itself: no barrier needed)
    }
    unsigned ncalls = 0;                                // residual evaluations of this orbit (measurement aid)
    auto rows = [&](double q, double P, double &r1, double &r2) {   // Kstar(1,:).alpha, Kstar(2,:).alpha
        r1 = 0.0; r2 = 0.0;
        ++ncalls;
        if (staged) {
            for (int i = 0, j = me * TT + threadIdx.x; j < a.n0; ++i, j += S * TT) {
                const int l = i * TT + threadIdx.x;
                double kxx, kxy, kyy;
                pair_eval<FAM, false>(st[0][l], st[1][l], q, P, a.kc, kxx, kxy, kyy);
                const double a1 = st[2][l], a2 = st[3][l];
                r1 += kxx * a1 + kxy * a2;
                r2 += kxy * a1 + kyy * a2;
            }
        } else {
            for (int j = me * TT + threadIdx.x; j < a.n0; j += S * TT) {
                double kxx, kxy, kyy;
                pair_eval<FAM, false>(a.xtr[j], a.ytr[j], q, P, a.kc, kxx, kxy, kyy);
                const double a1 = a.alpha[j], a2 = a.alpha[a.n0 + j];
                r1 += kxx * a1 + kxy * a2;
                r2 += kxy * a1 + kyy * a2;
            }
        }
        team_sum2(r1, r2);
    };
    auto guess = [&](double q, double p) {
        double r = 0.0, z = 0.0;
        if (staged) {
            for (int i = 0, j = me * TT + threadIdx.x; j < a.n0p; ++i, j += S * TT) {
                const int l = i * TT + threadIdx.x;
                r += a.kcp.sig * kern_eval<FAM, false>(st[4][l], st[5][l], q, p, a.kcp) * st[6][l];
            }
        } else {
            for (int j = me * TT + threadIdx.x; j < a.n0p; j += S * TT)
                r += a.kcp.sig * kern_eval<FAM, false>(a.xtrp[j], a.ytrp[j], q, p, a.kcp) * a.alphap[j];
        }
        team_sum2(r, z);
        return r;
    };
    double q = a.Q0[k], p = a.P0[k], pd = p;
    if (threadIdx.x == 0 && me == 0) {
        a.qmap[k] = q;
        a.pmap[k] = p;
        if (a.pdiff) a.pdiff[k] = pd;
    }
    const double nan = __builtin_nan("");
    const double twopi = 6.283185307179586477;
    for (int i = 0; i + 1 < a.nm; ++i) {
        double qn = nan, pn = nan, pdn = nan;
        // some team of this call has given up (a.err): every member of every team sees it at its next step and writes NaN for
        // the rest of its orbit instead of waiting for partners that have diverged
        if (S > 1 && !lost && __hip_atomic_load((__attribute__((address_space(1))) int *)a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) lost = true;
        if (!lost && !(q != q) && !(p != p)) {             // NaN = lost orbit stays lost (func.py:231-232)
            double r1, r2, Praw = nan;
            bool have_r2 = false;
            if (a.mode & SGPR_MAP_EXPLICIT) {
                // explicit map (01_pendulum/explicit/func_expl.py:106-119, 04_standard_map/func.py:174-179):
                // P = p - Kstar(1,:).alpha at (q, p), no implicit equation
                rows(q, p, r1, r2);
                Praw = p - r1;
            } else {
                double P0 = guess(q, p);
                rows(q, P0, r1, r2);
                double f0 = r1 - p + P0;
                double P1 = P0 - f0;                            // f'(P) ~ 1 near the identity map
                rows(q, P1, r1, r2);
                double f1 = r1 - p + P1;
                for (int it = 0; it < a.maxiter; ++it) {        // secant; every quantity is block-uniform
                    if (!(fabs(P1 - P0) > a.tol * fmax(1.0, fabs(P1))) || !(f1 == f1)) break;
                    const double d = f1 - f0;
                    if (d == 0.0) break;
                    const double Pn = P1 - f1 * (P1 - P0) / d;
                    P0 = P1; f0 = f1; P1 = Pn;
                    rows(q, P1, r1, r2);
                    f1 = r1 - p + P1;
                }
                if ((f1 == f1) && fabs(f1) <= 1e-8 * fmax(1.0, fabs(p))) {
                    Praw = P1;
                    have_r2 = true;                              // r2 belongs to (q, P1)
                }
            }
            // 05_tokamak/SympGPR/func.py:190-211, sympgpr.f90:128-177: an orbit whose new momentum is negative has left the
            // plasma -- lost from this step on (the flux-surface half of that test needs the out-of-scope fieldlines module and
            // stays with the caller: examples/tokamak.py)
            if ((a.mode & SGPR_MAP_LOSS_NEGP) && Praw < 0.0) Praw = nan;
            if (Praw == Praw) {
                pdn = pd + (Praw - p);                           // unwrapped momentum (04_standard_map/func.py:234)
                pn = Praw;
                if (a.mode & SGPR_MAP_WRAP_P) pn -= twopi * floor(pn / twopi);
                if (!have_r2 || pn != Praw) rows(q, pn, r1, r2);
                qn = r2 + q;                                     // Eq. (43)
                if (a.mode & SGPR_MAP_WRAP_Q) qn -= twopi * floor(qn / twopi);
            }
        }
        q = qn; p = pn; pd = pdn;
        if (threadIdx.x == 0 && me == 0) {
            a.qmap[(size_t)(i + 1) * a.ntest + k] = q;
            a.pmap[(size_t)(i + 1) * a.ntest + k] = p;
            if (a.pdiff) a.pdiff[(size_t)(i + 1) * a.ntest + k] = pd;
        }
    }
    if (threadIdx.x == 0 && me == 0) atomicAdd((unsigned *)(a.err + 1), ncalls);
}

template <typename F>
int dispatch_family(int family, F &&f)
{
    switch (family) {
    case SGPR_FAM_A: return f(std::integral_constant<int, SGPR_FAM_A>());
    case SGPR_FAM_B: return f(std::integral_constant<int, SGPR_FAM_B>());
    case SGPR_FAM_C: return f(std::integral_constant<int, SGPR_FAM_C>());
    case SGPR_FAM_D: return f(std::integral_constant<int, SGPR_FAM_D>());
    case SGPR_FAM_USER: return f(std::integral_constant<int, SGPR_FAM_USER>());
    }
    set_error("unknown kernel family");
    return SGPR_E_ARG;
}

}  // namespace

bool family_has_p(int family) { return family == SGPR_FAM_D || (family == SGPR_FAM_USER && gen::user_has_p); }

int make_kconst(int family, const double *hyp, int nhyp, KConst *out)
{
    const bool has_p = family_has_p(family);
    const int need = has_p ? 4 : 3;
    if (family < SGPR_FAM_A || family > SGPR_FAM_USER || !hyp || nhyp != need) {
        set_error("hyp must hold (lx, ly, sig) -- (lx, ly, p, sig) for family D");
        return SGPR_E_ARG;
    }
    KConst k{};
    k.lx = hyp[0];
    k.ly = hyp[1];
    k.p = has_p ? hyp[2] : 0.0;
    k.sig = hyp[nhyp - 1];
    k.lx2 = k.lx * k.lx;
    k.ly2 = k.ly * k.ly;
    k.inv_lx2 = 1.0 / k.lx2;
    k.inv_ly2 = 1.0 / k.ly2;
    const double pp = family == SGPR_FAM_D ? k.p * k.p : (family == SGPR_FAM_C ? 1.0 : 0.25);
    const double pm = family == SGPR_FAM_D ? k.p : (family == SGPR_FAM_C ? 1.0 : 0.5);
    k.cxx = k.sig * pp / (k.lx2 * k.lx2);
    k.cyy = k.sig / (k.ly2 * k.ly2);
    k.cxy = -k.sig * pm / (k.lx2 * k.ly2);
    k.hscale = family == SGPR_FAM_D ? k.p : 0.5;
    k.inv_lx = 1.0 / k.lx;
    k.inv_ly = 1.0 / k.ly;
    k.inv_lx3 = 1.0 / (k.lx2 * k.lx);
    k.inv_ly3 = 1.0 / (k.ly2 * k.ly);
    k.gxx = k.sig * pp;
    *out = k;
    return 0;
}

int make_kconst_l(int family, const double *l, int nl, KConst *out)
{
    double h[4];
    const int need = family_has_p(family) ? 3 : 2;
    if (!l || nl != need) {
        set_error("l must hold (lx, ly) -- (lx, ly, p) for family D");
        return SGPR_E_ARG;
    }
    for (int i = 0; i < nl; ++i) h[i] = l[i];
    h[nl] = 1.0;
    return make_kconst(family, h, nl + 1, out);
}

int gram_pairs(int family, int mi, int mj, const double *xb, const double *yb, const double *xa,
               const double *ya, const KConst &kc, double *qq, double *Pq, double *qP, double *PP,
               size_t ld, long diag_off, double noise, unsigned flags, hipStream_t st)
{
    if (mi < 0 || mj < 0) { set_error("negative extent"); return SGPR_E_ARG; }
    if (mi == 0 || mj == 0) return 0;
    unsigned parts = flags & SGPR_G_ALL;
    if (!qq) parts &= ~SGPR_G_QQ;
    if (!Pq) parts &= ~SGPR_G_PQ;
    if (!qP) parts &= ~SGPR_G_QP;
    if (!PP) parts &= ~SGPR_G_PP;
    if (!parts) return 0;
    if (ld < (size_t)mi) { set_error("ld smaller than the tile's row count"); return SGPR_E_ARG; }
    GramArgs a;
    a.mi = mi; a.mj = mj; a.xb = xb; a.yb = yb; a.xa = xa; a.ya = ya;
    a.dst[0] = qq; a.dst[1] = Pq; a.dst[2] = qP; a.dst[3] = PP;
    a.ld = ld; a.diag_off = diag_off; a.noise = noise;
    a.flags = parts | (flags & SGPR_G_LOWER);
    a.kc = kc;
    const dim3 grid((mi + TI - 1) / TI, (mj + TJ - 1) / TJ);
    if (grid.y > 65535) { set_error("too many pair columns for one launch"); return SGPR_E_ARG; }
    const bool ocml = flags & SGPR_G_OCML;
    const int deriv = (flags & SGPR_G_DLX) ? DERIV_LX : ((flags & SGPR_G_DLY) ? DERIV_LY : DERIV_NONE);
    return dispatch_family(family, [&](auto fam) {
        constexpr int F = decltype(fam)::value;
        if (deriv == DERIV_LX)      hipLaunchKernelGGL((gram_pairs_kernel<F, false, DERIV_LX>), grid, dim3(GT), 0, st, a);
        else if (deriv == DERIV_LY) hipLaunchKernelGGL((gram_pairs_kernel<F, false, DERIV_LY>), grid, dim3(GT), 0, st, a);
        else if (ocml) hipLaunchKernelGGL((gram_pairs_kernel<F, true, DERIV_NONE>), grid, dim3(GT), 0, st, a);
        else           hipLaunchKernelGGL((gram_pairs_kernel<F, false, DERIV_NONE>), grid, dim3(GT), 0, st, a);
        SGPR_CHECK_LAUNCH();
        return 0;
    });
}

int gram_reg(int family, int mi, int mj, const double *xb, const double *yb, const double *xa,
             const double *ya, const KConst &kc, double *G, size_t ld, long diag_off, double noise,
             hipStream_t st, int deriv)
{
    if (mi < 0 || mj < 0) { set_error("negative extent"); return SGPR_E_ARG; }
    if (mi == 0 || mj == 0) return 0;
    if (ld < (size_t)mi) { set_error("ld smaller than the tile's row count"); return SGPR_E_ARG; }
    RegArgs a{mi, mj, xb, yb, xa, ya, G, ld, diag_off, noise, kc};
    const dim3 grid((mi + TI - 1) / TI, (mj + TJ - 1) / TJ);
    if (grid.y > 65535) { set_error("too many pair columns for one launch"); return SGPR_E_ARG; }
    return dispatch_family(family, [&](auto fam) {
        constexpr int F = decltype(fam)::value;
        if (deriv == DERIV_LX)      hipLaunchKernelGGL((gram_reg_kernel<F, false, DERIV_LX>), grid, dim3(GT), 0, st, a);
        else if (deriv == DERIV_LY) hipLaunchKernelGGL((gram_reg_kernel<F, false, DERIV_LY>), grid, dim3(GT), 0, st, a);
        else                        hipLaunchKernelGGL((gram_reg_kernel<F, false, DERIV_NONE>), grid, dim3(GT), 0, st, a);
        SGPR_CHECK_LAUNCH();
        return 0;
    });
}

int kernel_eval(int family, int which, int m, const double *xa, const double *ya, const double *xb,
                const double *yb, const KConst &kc, double *out, hipStream_t st)
{
    if (m <= 0) return 0;
    if (which >= SGPR_K_DX && which <= SGPR_K_DXDY0DY0)
        return dispatch_family(family, [&](auto fam) {
            constexpr int F = decltype(fam)::value;
            hipLaunchKernelGGL((kernel_extra_kernel<F>), dim3((m + 255) / 256), dim3(256), 0, st, which, m, xa, ya,
                               xb, yb, kc, out);
            SGPR_CHECK_LAUNCH();
            return 0;
        });
    const int deriv = which >> 2;
    which &= 3;
    if (deriv < 0 || deriv > 2) { set_error("unknown kernel function"); return SGPR_E_ARG; }
    return dispatch_family(family, [&](auto fam) {
        constexpr int F = decltype(fam)::value;
        const dim3 grid((m + 255) / 256);
        if (deriv == DERIV_LX)      hipLaunchKernelGGL((kernel_eval_kernel<F, DERIV_LX>), grid, dim3(256), 0, st, which, m, xa, ya, xb, yb, kc, out);
        else if (deriv == DERIV_LY) hipLaunchKernelGGL((kernel_eval_kernel<F, DERIV_LY>), grid, dim3(256), 0, st, which, m, xa, ya, xb, yb, kc, out);
        else                        hipLaunchKernelGGL((kernel_eval_kernel<F, DERIV_NONE>), grid, dim3(256), 0, st, which, m, xa, ya, xb, yb, kc, out);
        SGPR_CHECK_LAUNCH();
        return 0;
    });
}

int predict_rows(int family, int m, const double *q, const double *P, int n0, const double *xtr,
                 const double *ytr, const KConst &kc, const double *alpha, double *out_p,
                 double *out_q, hipStream_t st)
{
    if (m <= 0) return 0;
    return dispatch_family(family, [&](auto fam) {
        constexpr int F = decltype(fam)::value;
        hipLaunchKernelGGL((predict_rows_kernel<F>), dim3(m), dim3(GT), 0, st, n0, q, P, xtr, ytr,
                           kc, alpha, out_p, out_q);
        SGPR_CHECK_LAUNCH();
        return 0;
    });
}

// Workgroups per orbit: TWO 256-thread members per CU (512 slots) shared out among the ntest orbits, at most 16 per orbit, and no
// more than the training set can feed with a round of 256 points each -- the drivers' own sizes (20 - 80 points) keep one
// workgroup per orbit.  Two members of DIFFERENT orbits per CU is the point: a residual is ~3.5 us of VALU work per CU and ~5 us
// of reduction + exchange with the rest of the team, and with one 512-thread member per CU (the first form: 81 G pair
// evaluations per second at N0 = 16 384, Ntest = 37) the CU idles through the exchange; with two the other orbit computes.
// Every member of a team has to be resident at once (they wait for each other): 512 workgroups of 4 waves fit the chip at
// <= 170 VGPRs (3 waves per SIMD; family A, the largest, has 136).
int applymap_team(int ntest, int n0)
{
    // resident workgroups the device can hold: two of these per CU (queried once per device; 256 CUs -> 512)
    static const int slots = [] {
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) {
            (void)hipGetLastError();
            return 0;                                    // unknown device: no teams (S = 1 needs no co-residency)
        }
        return 2 * ncu;
    }();
    int S = ntest > 0 ? slots / ntest : 1;
    S = std::min(S, (n0 + MAP_TEAM_T - 1) / MAP_TEAM_T);
    return std::max(1, std::min(S, MAP_TEAM_MAX));
}
size_t applymap_team_ws(int ntest, int n0)
{
    return ((size_t)ntest * applymap_team(ntest, n0) * 2 * 4 + 2) * sizeof(unsigned long long);    // granules + the error word
}

// team_ws: applymap_team_ws(ntest, n0) bytes of device scratch (cleared here)
int applymap(int family, int mode, int nm, int ntest, int n0, const double *xtr, const double *ytr,
             const KConst &kc, const double *alpha, int n0p, const double *xtrp, const double *ytrp,
             const KConst &kcp, const double *alphap, const double *Q0, const double *P0, double *qmap,
             double *pmap, double *pdiff, void *team_ws, hipStream_t st)
{
    if (nm <= 0 || ntest <= 0) return 0;
    const int S = applymap_team(ntest, n0);
    SGPR_HIP(hipMemsetAsync(team_ws, 0, applymap_team_ws(ntest, n0), st));
    unsigned long long *tw = static_cast<unsigned long long *>(team_ws);
    int *err = reinterpret_cast<int *>(tw + (size_t)ntest * S * 2 * 4);
    MapArgs a{nm, ntest, n0, n0p, mode, 60, S, tw, err, 1e-13, xtr, ytr, alpha, xtrp, ytrp, alphap, Q0, P0, qmap, pmap, pdiff, kc, kcp};
    return dispatch_family(family, [&](auto fam) {
        constexpr int F = decltype(fam)::value;
        if (S == 1) hipLaunchKernelGGL((applymap_kernel<F, GT>), dim3(ntest), dim3(GT), 0, st, a);
        else        hipLaunchKernelGGL((applymap_kernel<F, MAP_TEAM_T>), dim3(ntest * S), dim3(MAP_TEAM_T), 0, st, a);
        SGPR_CHECK_LAUNCH();
        return 0;
    });
}
static std::atomic<unsigned> g_last_map_calls{0};
unsigned applymap_last_calls() { return g_last_map_calls.load(); }    // measurement aid (libsympgpr_probe.so): K*-row evaluations of the last map

// after the stream has been waited for: did a team member give up on another (never a property of the data)?
int applymap_status(const void *team_ws, int ntest, int n0)
{
    const unsigned long long *tw = static_cast<const unsigned long long *>(team_ws);
    int hh[2] = {0, 0};
    SGPR_HIP(hipMemcpy(hh, tw + (size_t)ntest * applymap_team(ntest, n0) * 2 * 4, sizeof(hh), hipMemcpyDeviceToHost));
    g_last_map_calls.store((unsigned)hh[1]);
    const int h = hh[0];
    if (h) { set_error("applymap: a workgroup of an orbit's team did not answer in time"); return SGPR_E_HIP; }
    return 0;
}

int predict_reg(int family, int m, const double *q, const double *P, int n0, const double *xtr,
                const double *ytr, const KConst &kc, const double *alpha, double *out,
                hipStream_t st)
{
    if (m <= 0) return 0;
    return dispatch_family(family, [&](auto fam) {
        constexpr int F = decltype(fam)::value;
        hipLaunchKernelGGL((predict_reg_kernel<F>), dim3(m), dim3(GT), 0, st, n0, q, P, xtr, ytr,
                           kc, alpha, out);
        SGPR_CHECK_LAUNCH();
        return 0;
    });
}

}  // namespace sgpr

// cholq.hip -- plan builder and worker kernel of the task-queue Cholesky (see cholq.h).
//
// The worker grid is persistent for the whole factorisation of one block: a workgroup draws a ticket, reads
// the task behind it, waits until the task's inputs carry the version it expects, runs it on the fp64 matrix
// cores with the LDS-DMA body of the grid-wide kernel (gemm_tile.h, one call site), publishes the new version
// and draws the next ticket.  Forward progress needs no co-residency of the workers: the list is ordered so
// that every dependency of a task has a smaller ticket, a workgroup works its tickets in increasing order, so
// the smallest unfinished ticket can always run.  What the workers DO rely on is the chain of diagonal blocks
// making progress beside them (panel kernels on the CUs this grid leaves free, chol.hip); every spin is bounded
// and an abort word stops the whole grid when one of them runs out.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

#include "cholq.h"
#include "gemm_tile.h"

namespace sgpr {
namespace cholq {

namespace {

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
int q_min() { static const int v = env_int("SGPR_Q_MIN", 2048); return v; }
int q_max() { static const int v = std::min(env_int("SGPR_Q_MAX", 57344), MAX_ORDER); return v; }
bool q_on() { static const int v = env_int("SGPR_POTRF_Q", 1); return v != 0; }

size_t pad256(size_t b) { return (b + 255) / 256 * 256; }

// lower-triangle elements of tile (i, j): rows [256 i, +256), columns [128 j, +128), row >= column
double tile_lower_elems(int i, int j)
{
    const long r0 = (long)TM * i, c0 = (long)TN * j;
    if (c0 + TN - 1 <= r0) return (double)TM * TN;
    double e = 0.0;
    for (long c = c0; c < c0 + TN; ++c) {
        const long first = std::max(r0, c);
        if (first < r0 + TM) e += (double)(r0 + TM - first);
    }
    return e;
}

std::mutex g_plan_mu;
std::map<std::pair<int, int>, Plan *> g_plans;

unsigned long long *g_trace = nullptr;     // diagnostics only
size_t g_trace_cap = 0;

}  // namespace

bool eligible(int n) { return q_on() && n >= q_min() && n <= q_max() && n % TM == 0; }

std::vector<int> default_starts(int n)
{
    // Panel widths by what is left (all multiples of 256).  Wide panels while the trailing update is long enough
    // to hide the next panel's chain and rows-below solves (k = 1024 tiles run ~13 % faster than k = 512 ones),
    // narrower ones towards the end, and a ramp 256 -> 512 -> 1024 at the start: nothing can run beside the first
    // panel, so it is kept short.
    static const int w0 = std::max(256, env_int("SGPR_Q_W0", 256) / 256 * 256);
    static const int t0 = env_int("SGPR_Q_T0", 0);
    static const int t1 = env_int("SGPR_Q_T1", 6144);
    static const int t2 = env_int("SGPR_Q_T2", 2048);
    static const int wcap = std::max(256, env_int("SGPR_Q_WMAX", 2048) / 256 * 256);
    const int t0_eff = t0 > 0 ? t0 : (n >= 24576 ? 12288 : 1 << 30);
    std::vector<int> s;
    int pos = 0, k = 0;
    while (pos < n) {
        s.push_back(pos);
        const int rem = n - pos;
        int w = rem > t0_eff ? 2048 : (rem > t1 ? 1024 : (rem > t2 ? 512 : 256));
        w = std::min(w, wcap);
        w = std::min(w, w0 << std::min(k, 3));
        w = std::min(w, rem);
        pos += w;
        ++k;
    }
    s.push_back(n);
    return s;
}

int build_plan(int n, const std::vector<int> &starts, int nworkers, Plan &out)
{
    if (n <= 0 || n % TM != 0 || n > MAX_ORDER || starts.size() < 2 || starts.front() != 0 || starts.back() != n) {
        set_error("cholq: bad plan request");
        return SGPR_E_ARG;
    }
    const int nblk = (int)starts.size() - 1;
    for (int k = 0; k < nblk; ++k) {
        const int w = starts[k + 1] - starts[k];
        if (w <= 0 || w % TM != 0 || w > 2048) { set_error("cholq: panel widths must be multiples of 256, at most 2048"); return SGPR_E_ARG; }
    }
    if (nblk > 511) { set_error("cholq: too many panels"); return SGPR_E_ARG; }
    out = Plan();
    out.n = n; out.nblk = nblk; out.starts = starts; out.nworkers = nworkers;
    const int tm = n / TM;
    auto lower = [](int i, int j) { return TM * i + TM - 1 >= TN * j; };
    std::vector<unsigned> &tasks = out.tasks;
    double flop = 0.0;
    // rows of the NEXT diagonal block are solved by the panel kernel itself, in step with its chain (chol.hip: the
    // strips below the diagonal block): they are on the critical path panel -> update of the next diagonal block -> panel
    auto emit_t = [&](int k) {
        const double w = starts[k + 1] - starts[k];
        const int first = (k + 2 <= nblk) ? starts[k + 2] / TM : tm;
        for (int i = first; i < tm; ++i) {
            tasks.push_back(pack(TASK_T, k, i, 0));
            flop += (double)TM * w * w;
        }
    };
    emit_t(0);
    std::vector<unsigned> rest;
    for (int k = 0; k + 1 < nblk; ++k) {
        const int w = starts[k + 1] - starts[k];
        out.wmax = std::max(out.wmax, w);
        // block column k+1 first, rows ascending: the diagonal block of the next panel (its chain waits for it),
        // then the rows below (the next panel's solves wait for those)
        for (int i = starts[k + 1] / TM; i < tm; ++i)
            for (int j = starts[k + 1] / TN; j < starts[k + 2] / TN; ++j)
                if (lower(i, j)) { tasks.push_back(pack(TASK_U, k, i, j)); flop += 2.0 * w * tile_lower_elems(i, j); }
        // the other block columns in order, row tile by row tile inside each (one row panel, all its columns)
        rest.clear();
        for (int kk = k + 2; kk < nblk; ++kk)
            for (int i = starts[kk] / TM; i < tm; ++i)
                for (int j = starts[kk] / TN; j < starts[kk + 1] / TN; ++j)
                    if (lower(i, j)) { rest.push_back(pack(TASK_U, k, i, j)); flop += 2.0 * w * tile_lower_elems(i, j); }
        // the next panel's solves go in where its chain is expected to have finished: earlier and the workers that
        // draw them wait, later and the next step's first updates wait for them
        const int wn = (starts[k + 2] - starts[k + 1]) / LEAF;
        const double chain_us = 75.0 * wn + 60.0, tile_us = 3.56 * w / 16.0 + 20.0;
        const size_t pos = std::min(rest.size(), (size_t)std::ceil(chain_us / tile_us * std::max(nworkers, 1)));
        tasks.insert(tasks.end(), rest.begin(), rest.begin() + pos);
        emit_t(k + 1);
        tasks.insert(tasks.end(), rest.begin() + pos, rest.end());
    }
    out.wmax = std::max(out.wmax, starts[nblk] - starts[nblk - 1]);
    out.flop = flop;
    return 0;
}

const Plan *get_plan(int n, int nworkers)
{
    std::lock_guard<std::mutex> lock(g_plan_mu);
    const auto key = std::make_pair(n, nworkers);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    Plan *p = new Plan();
    if (build_plan(n, default_starts(n), nworkers, *p)) { delete p; return nullptr; }
    // page-locked image for the asynchronous upload: [starts | tasks]
    const size_t words = (size_t)(p->nblk + 1) + p->tasks.size();
    void *pin = nullptr;
    if (hipHostMalloc(&pin, words * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_error("cholq: hipHostMalloc failed");
        delete p;
        return nullptr;
    }
    p->pinned = static_cast<unsigned *>(pin);
    for (int k = 0; k <= p->nblk; ++k) p->pinned[k] = (unsigned)p->starts[k];
    std::copy(p->tasks.begin(), p->tasks.end(), p->pinned + p->nblk + 1);
    g_plans[key] = p;      // plans live for the life of the process (a few per order in use)
    return p;
}

// Upper bound of the queue's workspace for any block of order <= n: with 256-wide panels everywhere step k
// (trailing order M tiles of 256) has M (M + 1) update tiles and M solves.
size_t ws_bytes(int n)
{
    if (!q_on()) return 0;
    const int nq = std::min(n, q_max()) / TM * TM;
    if (nq < q_min()) return 0;
    const size_t tm = (size_t)nq / TM, tn = (size_t)nq / TN;
    size_t ntasks = 0;
    for (size_t M = 1; M <= tm; ++M) ntasks += M * (M + 1) + M;
    return 256 + pad256(tm * tn * 4) + pad256(tn * 4) + pad256((tm + 1) * 4) + pad256(ntasks * 4) + 256;
}

Ws carve(void *base, int n)
{
    Ws w{};
    const size_t tm = (size_t)n / TM, tn = (size_t)n / TN;
    char *p = reinterpret_cast<char *>(((uintptr_t)base + 255) / 256 * 256);
    w.qs = reinterpret_cast<int *>(p);            p += 256;
    w.ver = reinterpret_cast<int *>(p);           p += pad256(tm * tn * 4);
    w.tver = reinterpret_cast<int *>(p);          p += pad256(tn * 4);
    w.zero_bytes = (size_t)(p - reinterpret_cast<char *>(w.qs));
    w.pstart = reinterpret_cast<int *>(p);        p += pad256((tm + 1) * 4);
    w.tasks = reinterpret_cast<unsigned *>(p);
    return w;
}

unsigned long long *trace_panel_base(int ntasks)
{
    return (g_trace && (size_t)ntasks <= g_trace_cap) ? g_trace + 4 * g_trace_cap + 2 * TRACE_WORKERS : nullptr;
}

namespace {
Ws g_last_ws{};
int g_last_n = 0, g_last_ntasks = 0;
}
void remember(const Ws &w, int n, int ntasks) { g_last_ws = w; g_last_n = n; g_last_ntasks = ntasks; }

int postmortem(bool always)
{
    if (!g_last_n) return 0;
    const int n = g_last_n, tn = n / TN, tm = n / TM;
    int h[32];
    SGPR_HIP(hipMemcpy(h, g_last_ws.qs, sizeof(h), hipMemcpyDeviceToHost));
    if (!h[1] && !always) return 0;
    fprintf(stderr, "cholq n=%d %s: head %d of %d tasks; worker: set %d task %08x (type %u k %u i %u j %u) ticket %d short-mask %x head-then %d | "
            "panel: set %d block %d idx0 %d cnt %d need %d\n", n, h[1] ? "GAVE UP" : "state", h[0], g_last_ntasks, h[2], (unsigned)h[4], (unsigned)h[4] >> 30,
            ((unsigned)h[4] >> 21) & 511u, ((unsigned)h[4] >> 11) & 1023u, (unsigned)h[4] & 2047u, h[6], (unsigned)h[5], h[7],
            h[8], h[9], h[10], h[11], h[12]);
    std::vector<int> tv((size_t)tn), vv((size_t)tm * tn);
    SGPR_HIP(hipMemcpy(tv.data(), g_last_ws.tver, tv.size() * 4, hipMemcpyDeviceToHost));
    SGPR_HIP(hipMemcpy(vv.data(), g_last_ws.ver, vv.size() * 4, hipMemcpyDeviceToHost));
    fprintf(stderr, "  tver:");
    for (size_t r = 0; r < tv.size() && r < 160; ++r) fprintf(stderr, " %d", tv[r]);
    fprintf(stderr, "\n  ver (row tiles x column tiles, mod 36):\n");
    for (int i = 0; i < tm && i < 48; ++i) {
        fprintf(stderr, "   ");
        for (int j = 0; j < tn && j < 96; ++j) fprintf(stderr, "%c", "0123456789abcdefghijklmnopqrstuvwxyz"[vv[(size_t)i * tn + j] % 36]);
        fprintf(stderr, "\n");
    }
    return h[1];
}

void set_trace(unsigned long long *dev_buf, size_t capacity_tasks)
{
    g_trace = dev_buf;
    g_trace_cap = capacity_tasks;
}

namespace {

using namespace tile;
typedef __attribute__((address_space(1))) int gint;

struct QArgs {
    double *A;                  // the block being factored
    size_t lda;
    int n, tn;
    const unsigned *tasks;
    int ntasks;
    const int *pstart;
    int *qs, *ver, *tver;
    const int *flags;           // the panel kernel's hand-off words of this block: panel starting at leaf column t at flags + t * pstride
    int pstride;
    const double *inv;          // leaf inverses of this block
    int *dinfo;
    unsigned long long *trace;  // 4 words per ticket, or null
    unsigned long long *census; // 2 words per worker workgroup, or null
};

constexpr int QT = 512;
constexpr unsigned Q_SPIN_LIMIT = 3u << 20;     // ~ 3 s

__global__ __launch_bounds__(QT, 2) void chol_queue_kernel(const QArgs a)
{
    constexpr int LDA_S = TM + PAD, LDB_S = TN + PAD;
    __shared__ double smem[2 * BK * (LDA_S + LDB_S)];
    __shared__ int sh[4];
    const int tid = threadIdx.x;
    if (tid == 0) sh[0] = atomicAdd(a.qs, 1);
    if (a.census && tid == 0 && blockIdx.x < TRACE_WORKERS) {
        a.census[2 * blockIdx.x] = ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) |
                                   (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        a.census[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    int t = sh[0];
    while (t < a.ntasks) {
        const unsigned tk = a.tasks[t];
        const int type = (int)(tk >> 30), k = (int)((tk >> 21) & 511u), i = (int)((tk >> 11) & 1023u), j = (int)(tk & 2047u);
        const int s0 = a.pstart[k], w = a.pstart[k + 1] - s0;
        const int W = w / LEAF, j0 = s0 / TN;
        const bool tr = a.trace != nullptr && tid == 0;
        if (tr) a.trace[4 * (size_t)t] = __builtin_amdgcn_s_memrealtime();
        // the ticket after this one is drawn now and used at the bottom: its round trip hides under the task
        int tnext = 0;
        if (tid == 0) tnext = atomicAdd(a.qs, 1);
        // ---- inputs ready?  One lane per word, relaxed polls, then ONE agent-scope acquire for the workgroup
        if (tid < 64) {
            const int *p = nullptr;
            int need = 0;
            if (type == TASK_U) {
                if (tid == 0)      { p = a.ver + (size_t)i * a.tn + j; need = k; }
                else if (tid == 1) { p = a.tver + 2 * i;               need = k + 1; }
                else if (tid == 2) { p = a.tver + 2 * i + 1;           need = k + 1; }
                else if (tid == 3) { p = a.tver + j;                   need = k + 1; }
            } else {
                if (tid < W)          { p = a.ver + (size_t)i * a.tn + j0 + tid; need = k; }
                else if (tid < 2 * W) { p = a.flags + (size_t)j0 * a.pstride + 2 + (tid - W); need = 1; }   // I[c]: leaf c inverted
            }
            unsigned spins = 0;
            int ok = 1;
            for (;;) {
                const int v = p ? __hip_atomic_load((gint *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                if (__all(v >= need)) break;
                __builtin_amdgcn_s_sleep(8);
                if ((++spins & 63u) == 0) {
                    const int ab = __hip_atomic_load((gint *)(a.qs + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ab != 0 || spins > Q_SPIN_LIMIT) { ok = 0; break; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!ok && spins > Q_SPIN_LIMIT) {
                // post-mortem (SGPR_Q_DEBUG=1 prints it): the first task that gave up, and which of its words were short
                const int v = p ? __hip_atomic_load((gint *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                const unsigned long long short_mask = __ballot(v < need);
                if (tid == 0 && atomicCAS(a.qs + 2, 0, 1) == 0) {
                    a.qs[4] = (int)tk; a.qs[5] = (int)(short_mask & 0xffffffffu); a.qs[6] = t; a.qs[7] = a.qs[0];
                }
            }
            if (tid == 0) {
                if (!ok) {
                    if (spins > Q_SPIN_LIMIT) atomicCAS(a.dinfo, 0, POTRF_HANDOFF_TIMEOUT);
                    __hip_atomic_store((gint *)(a.qs + 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sh[1] = ok;
            }
        }
        __syncthreads();
        if (!sh[1]) break;                       // workgroup-uniform: the factorisation has been given up
        if (tr) a.trace[4 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
        // ---- the task's products (one call site of the k-loop body)
        const int nprod = (type == TASK_U) ? 1 : 2 * W - 1;
        for (int p = 0; p < nprod; ++p) {
            GemmArgs g{};
            g.m = a.n; g.n = a.n; g.lda = a.lda; g.ldc = a.lda; g.stamps = nullptr;
            int trow = 0, tcol = 0;
            if (type == TASK_U) {
                g.A = a.A + (size_t)s0 * a.lda; g.B = g.A; g.ldb = a.lda;
                g.C = a.A; g.k = w; g.alpha = -1.0; g.beta = 1.0;
                trow = i; tcol = j;
            } else {
                // leaf column c of the panel: p = 2c - 1 folds the solved columns 0..c-1 into it, p = 2c multiplies it
                // with inv(L_cc)^T in place (the workgroup owns full rows: every read of the tile precedes its stores)
                const int c = (p + 1) >> 1;
                double *X = a.A + (size_t)i * TM + (size_t)(s0 + c * LEAF) * a.lda;
                if (p & 1) {
                    g.A = a.A + (size_t)i * TM + (size_t)s0 * a.lda;
                    g.B = a.A + (size_t)(s0 + c * LEAF) + (size_t)s0 * a.lda; g.ldb = a.lda;
                    g.k = c * LEAF; g.alpha = -1.0; g.beta = 1.0;
                } else {
                    g.A = X;
                    g.B = a.inv + (size_t)(j0 + c) * LEAF * LEAF; g.ldb = LEAF;
                    g.k = LEAF; g.alpha = 1.0; g.beta = 0.0;
                }
                g.C = X;
            }
            if (p > 0) {                         // this workgroup's own stores of the previous product feed this one
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            gemm_body_dma<TM, TN, 2>(g, smem, trow, tcol);
        }
        // ---- publish: every wave drains its stores, barrier, one lane releases and bumps the version
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (type == TASK_U) {
                __hip_atomic_store((gint *)(a.ver + (size_t)i * a.tn + j), k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store((gint *)(a.tver + 2 * i), k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store((gint *)(a.tver + 2 * i + 1), k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tr) {
                a.trace[4 * (size_t)t + 2] = __builtin_amdgcn_s_memrealtime();
                a.trace[4 * (size_t)t + 3] = ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) | tk;
            }
            sh[0] = tnext;
        }
        __syncthreads();
        t = sh[0];
    }
}

}  // namespace

int prepare(const Plan &p, const Ws &w, hipStream_t st)
{
    SGPR_HIP(hipMemsetAsync(w.qs, 0, w.zero_bytes, st));
    SGPR_HIP(hipMemcpyAsync(w.pstart, p.pinned, (size_t)(p.nblk + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    SGPR_HIP(hipMemcpyAsync(w.tasks, p.pinned + p.nblk + 1, p.tasks.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
    return 0;
}

int launch_workers(const Plan &p, const Ws &w, double *A, size_t lda, const double *inv, const int *flags, int *dinfo,
                   int pflag_stride, hipStream_t st)
{
    if (p.tasks.empty()) return 0;
    QArgs a{};
    a.A = A; a.lda = lda; a.n = p.n; a.tn = p.n / TN;
    a.tasks = w.tasks; a.ntasks = (int)p.tasks.size(); a.pstart = w.pstart;
    a.qs = w.qs; a.ver = w.ver; a.tver = w.tver;
    a.flags = flags; a.pstride = pflag_stride; a.inv = inv; a.dinfo = dinfo;
    a.trace = (g_trace && p.tasks.size() <= g_trace_cap) ? g_trace : nullptr;
    a.census = a.trace ? g_trace + 4 * g_trace_cap : nullptr;
    const int grid = std::max(1, std::min(p.nworkers, (int)p.tasks.size()));
    hipLaunchKernelGGL(chol_queue_kernel, dim3(grid), dim3(QT), 0, st, a);
    SGPR_CHECK_LAUNCH();
    return 0;
}

}  // namespace cholq
}  // namespace sgpr

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
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <queue>

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
int q_min() { static const int v = env_int("SGPR_Q_MIN", 13312); return v; }
int q_max() { static const int v = std::min(env_int("SGPR_Q_MAX", 28672), MAX_ORDER); return v; }
// ON by default (SGPR_POTRF_Q=0 switches it off) for SGPR_Q_MIN <= n <= SGPR_Q_MAX, where it beats the look-ahead driver.
// A persistent grid that fills every CU can lose workgroups for a while when the platform switches the queues out and in
// (DESIGN.md section 3.9); the workers notice (nothing published anywhere for 3 ms), drain their kernel instance and the next
// instance carries on (Q_INSTANCES).  Whether the worker grid and the panel kernel CAN run side by side on this process's
// streams is not guessed from the environment any more (round 3 looked for a profiler's ROCPROF_COUNTER_COLLECTION): chol.hip
// tries it once per device with a pair of handshake kernels (queue_overlap_ok) -- counter collection, HIP_LAUNCH_BLOCKING,
// AMD_SERIALIZE_KERNEL, a debugger: whatever serialises dispatches fails that test and the look-ahead driver is used.
bool q_on()
{
    static const int v = env_int("SGPR_POTRF_Q", 1);
    return v != 0;
}

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
    // Uniform panels (multiples of 256).  The planner below gives far tiles several panels per update task, so the panel
    // width no longer sets the k of the bulk of the flop; it sets the chain: a 512-wide panel costs the panel kernel
    // ~0.4 ms (4 leaf columns + the in-panel updates of its strips), and the next one can start one k = 512 tile later.
    static const int w = std::min(2048, std::max(256, (int)tune("q_w", 512) / 256 * 256));
    std::vector<int> s;
    for (int pos = 0; pos < n; pos += w) s.push_back(pos);
    s.push_back(n);
    return s;
}

// ---- the planner ----------------------------------------------------------------------------------------------------
// A list scheduler run ahead of time on a cost model of the device (constants measured with tools/queue_trace.py):
// `nworkers` workers, the panel kernel's chain beside them.  Whenever a worker is free it gets the most urgent task that
// is ready -- urgency = the panel that will consume the tile, then the row -- and an update task always takes EVERYTHING
// that is available for its tile (all leaf columns whose rows of L are final, up to kcap): tiles next to the chain are
// updated eagerly in thin slices, tiles far from it pile up several panels and are updated with k = 1024 .. 2048 at the
// matrix cores' best rate, without anybody choosing a block size for them.  The tasks are emitted in the order the model
// starts them; the device runs them in that order from one ticket counter, so every input of a task is produced by a
// task with a smaller ticket (or by the panel kernel, whose own inputs are), whatever the real timing turns out to be.
namespace {

struct Model {
    // DELIBERATELY PESSIMISTIC (measured: fixed ~15 us, a leaf column 66 - 90 us): the device runs the list in the
    // planner's order, a workgroup that draws a task waits for its inputs, so a plan that is ahead of the real chain
    // turns into workers waiting in line, while a plan that is behind it only costs a little look-ahead.
    double kstep = 3.56;      // one k-step (16 columns) of the 256 x 128 body, us
    double fixed = 28.0;      // per product: the polls, first operand loads, epilogue, drain + release, the ticket
    double leaf = 72.0;       // chain: per leaf column ...
    double pair = 17.0;       // ... + per (strip, earlier column) pair of the panel: chain(W) = leaf W + pair W (W - 1) / 2
    double band0 = 40.0, band_pair = 12.0;   // the rows of the next diagonal block finish this long after the chain
    int kcap = 16;            // leaf columns per update task at most (k <= 2048)
};

struct Ev {
    double t;
    long seq;
    int kind, a, b, c;        // 0: chain a ends; 1: band of panel a; 2: update of (a, b) done, c = new version; 3: solve (panel a, row tile b) done
    bool operator>(const Ev &o) const { return t > o.t || (t == o.t && seq > o.seq); }
};
struct Cand {
    int p, type, i, j;        // key: consumer panel, solves before updates, row; j = column tile (update) or panel (solve)
    bool operator>(const Cand &o) const
    {
        if (p != o.p) return p > o.p;
        if (type != o.type) return type > o.type;
        if (i != o.i) return i > o.i;
        return j > o.j;
    }
};

}  // namespace

int default_nq(int n, const std::vector<int> &starts)
{
    static const int tail = std::max(0, (int)tune("q_tail", 0));
    const int nblk = (int)starts.size() - 1;
    int nq = nblk;
    while (nq > 1 && n - starts[nq - 1] <= tail) --nq;      // panel nq - 1 still has more than `tail` rows under and beside it
    return tail == 0 ? nblk : nq;
}

int build_plan(int n, const std::vector<int> &starts, int nworkers, Plan &out, int nq)
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
    if (nq < 0 || nq > nblk) nq = nblk;
    if (nq < 1) { set_error("cholq: the queue needs at least one panel"); return SGPR_E_ARG; }
    const int S = starts[nq];          // the queue's part: columns [0, S)
    Model M;
    // (12 up to n = 18432: 1 - 3 % faster there than 16 -- n = 14336 21.6 vs 22.2 ms, 16384 29.1 vs 29.5 -- the same above)
    M.kcap = std::max(1, std::min(16, (int)tune("q_kcap", n <= 18432 ? 12 : 16)));
    M.leaf = tune("q_leaf_us", M.leaf);
    M.pair = tune("q_pair_us", M.pair);
    M.fixed = tune("q_fixed_us", M.fixed);
    M.band0 = tune("q_band0_us", M.band0);
    out = Plan();
    out.n = n; out.nblk = nblk; out.starts = starts; out.nworkers = nworkers; out.nq = nq;
    for (int k = 0; k < nblk; ++k) out.wmax = std::max(out.wmax, starts[k + 1] - starts[k]);
    const int tm = n / TM, tn = n / TN;
    auto lower = [](int i, int j) { return TM * i + TM - 1 >= TN * j; };
    std::vector<int> pcol((size_t)tn), cap((size_t)tn);          // panel of column tile j; leaf columns it must have taken before its panel
    for (int k = 0; k < nblk; ++k)
        for (int j = starts[k] / TN; j < starts[k + 1] / TN; ++j) { pcol[j] = k; cap[j] = std::min(starts[k], S) / LEAF; }
    std::vector<int> ver((size_t)tm * tn, 0), rowfin((size_t)tn, 0);
    std::vector<char> busy((size_t)tm * tn, 0), inheap((size_t)tm * tn, 0);
    std::vector<char> tstate((size_t)nblk * tm, 3);              // 0 pending, 1 in heap, 2 issued, 3 none
    std::vector<int> tleft((size_t)nblk, 0);
    for (int k = 0; k < nq; ++k)
        for (int i = (k + 2 <= nblk ? starts[k + 2] / TM : tm); i < tm; ++i) { tstate[(size_t)k * tm + i] = 0; ++tleft[k]; }
    std::vector<char> chain_done((size_t)nblk, 0), band_done((size_t)nblk, 0);
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> events;
    std::priority_queue<Cand, std::vector<Cand>, std::greater<Cand>> cands;
    long seq = 0;
    double now = 0.0;
    int next_chain = 0;
    bool chain_running = false;
    int nfree = nworkers;
    double flop = 0.0;

    auto push_u = [&](int i, int j) {
        if (i < 0 || i >= tm || j < 0 || j >= tn || !lower(i, j)) return;
        const size_t q = (size_t)i * tn + j;
        if (inheap[q] || ver[q] >= cap[j]) return;
        inheap[q] = 1;
        cands.push(Cand{pcol[j], 1, i, j});
    };
    auto push_t = [&](int k, int i) {
        char &st = tstate[(size_t)k * tm + i];
        if (st != 0) return;
        st = 1;
        cands.push(Cand{k, 0, i, k});
    };
    auto set_rowfin = [&](int r, int v) {
        if (rowfin[r] >= v) return;
        rowfin[r] = v;
        const int i = r >> 1;
        for (int j = 0; j < tn && lower(i, j); ++j) push_u(i, j);      // the row tile of strip r ...
        for (int ii = r >> 1; ii < tm; ++ii) push_u(ii, r);           // ... and the column tile r (its rows are the B operand)
    };
    auto tiles_ready = [&](int i, int j0, int W, int need) {
        for (int c = 0; c < W; ++c) {
            const size_t q = (size_t)i * tn + j0 + c;
            if (ver[q] < need || busy[q]) return false;
        }
        return true;
    };
    auto try_chain = [&]() {
        if (chain_running || next_chain >= nq) return;
        const int k = next_chain, need = starts[k] / LEAF, j0 = starts[k] / TN, W = (starts[k + 1] - starts[k]) / LEAF;
        for (int g = 0; g < W; ++g) {
            const int vi = starts[k] / TM + (g >> 1);
            for (int c = 0; c <= g; ++c) {
                const size_t q = (size_t)vi * tn + j0 + c;
                if (ver[q] < need || busy[q]) return;
            }
        }
        chain_running = true;
        ++next_chain;
        events.push(Ev{now + M.leaf * W + M.pair * W * (W - 1) / 2.0, seq++, 0, k, 0, 0});
    };

    std::vector<unsigned> &tasks = out.tasks;
    try_chain();
    for (;;) {
        // hand the ready work to the free workers
        while (nfree > 0 && !cands.empty()) {
            const Cand cd = cands.top();
            cands.pop();
            if (cd.type == 1) {
                const int i = cd.i, j = cd.j;
                const size_t q = (size_t)i * tn + j;
                inheap[q] = 0;
                if (busy[q]) continue;
                const int a = ver[q];
                int b = std::min(std::min(rowfin[2 * i], rowfin[2 * i + 1]), std::min(rowfin[j], cap[j]));
                b = std::min(b, a + M.kcap);
                if (b <= a) continue;
                busy[q] = 1;
                --nfree;
                tasks.push_back(pack(TASK_U, 0, i, j));
                tasks.push_back(((unsigned)a << 16) | (unsigned)b);
                flop += 2.0 * LEAF * (b - a) * tile_lower_elems(i, j);
                events.push(Ev{now + M.kstep * 8.0 * (b - a) + M.fixed, seq++, 2, i, j, b});
            } else {
                const int k = cd.j, i = cd.i;
                char &st = tstate[(size_t)k * tm + i];
                st = 0;
                const int W = (starts[k + 1] - starts[k]) / LEAF;
                if (!chain_done[k] || !tiles_ready(i, starts[k] / TN, W, starts[k] / LEAF)) continue;
                st = 2;
                --tleft[k];
                for (int c = 0; c < W; ++c) busy[(size_t)i * tn + starts[k] / TN + c] = 1;
                --nfree;
                tasks.push_back(pack(TASK_T, k, i, 0));
                tasks.push_back(0u);
                const double w = starts[k + 1] - starts[k];
                flop += (double)TM * w * w;
                int ksteps = 8 * W;
                for (int c = 1; c < W; ++c) ksteps += 8 * c;
                events.push(Ev{now + M.kstep * ksteps + (2 * W - 1) * M.fixed * 0.5, seq++, 3, k, i, 0});
            }
        }
        if (events.empty()) break;
        const Ev ev = events.top();
        events.pop();
        now = ev.t;
        if (ev.kind == 0) {
            const int k = ev.a;
            chain_running = false;
            chain_done[k] = 1;
            const int W = (starts[k + 1] - starts[k]) / LEAF;
            for (int i = 0; i < tm; ++i)
                if (tstate[(size_t)k * tm + i] == 0) push_t(k, i);
            if (k + 1 < nblk) events.push(Ev{now + M.band0 + M.band_pair * W * (W - 1) / 2.0, seq++, 1, k, 0, 0});
            try_chain();
        } else if (ev.kind == 1) {
            // the rows of the next diagonal block: solved by the panel kernel once their own tiles were there
            const int k = ev.a, need = starts[k] / LEAF, j0 = starts[k] / TN, W = (starts[k + 1] - starts[k]) / LEAF;
            bool ready = true;
            for (int r = starts[k + 1] / LEAF; r < starts[k + 2] / LEAF && ready; ++r) ready = tiles_ready(r >> 1, j0, W, need);
            if (!ready) {
                events.push(Ev{now + 20.0, seq++, 1, k, 0, 0});
            } else {
                band_done[k] = 1;
                for (int r = starts[k + 1] / LEAF; r < starts[k + 2] / LEAF; ++r) set_rowfin(r, starts[k + 1] / LEAF);
            }
        } else if (ev.kind == 2) {
            const int i = ev.a, j = ev.b;
            const size_t q = (size_t)i * tn + j;
            ver[q] = ev.c;
            busy[q] = 0;
            ++nfree;
            push_u(i, j);
            if (ver[q] >= cap[j]) {
                const int k = pcol[j];
                if (chain_done[k] && tstate[(size_t)k * tm + i] == 0) push_t(k, i);
                try_chain();
            }
        } else {
            const int k = ev.a, i = ev.b, W = (starts[k + 1] - starts[k]) / LEAF;
            for (int c = 0; c < W; ++c) busy[(size_t)i * tn + starts[k] / TN + c] = 0;
            ++nfree;
            set_rowfin(2 * i, starts[k + 1] / LEAF);
            set_rowfin(2 * i + 1, starts[k + 1] / LEAF);
        }
    }
    // everything done?  (a planner bug would show here, never on the device)
    bool complete = next_chain == nq && !chain_running;
    for (int k = 0; k < nq && complete; ++k) complete = chain_done[k] && tleft[k] == 0 && (k + 1 >= nblk || band_done[k]);
    for (int i = 0; i < tm && complete; ++i)
        for (int j = 0; j < tn && complete; ++j)
            if (lower(i, j) && ver[(size_t)i * tn + j] < cap[j]) complete = false;
    if (!complete) { set_error("cholq: the planner left work undone"); return SGPR_E_HIP; }
    out.flop = flop;
    out.model_us = now;
    return 0;
}

const Plan *get_plan(int n, int nworkers)
{
    std::lock_guard<std::mutex> lock(g_plan_mu);
    const auto key = std::make_pair(n, nworkers);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    Plan *p = new Plan();
    const std::vector<int> st = default_starts(n);
    if (build_plan(n, st, nworkers, *p, default_nq(n, st)) || p->tasks.size() / 2 > max_tasks(n)) { delete p; g_plans[key] = nullptr; return nullptr; }
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

// Room for the task list of any block of order n: four times the tile count of a right-looking schedule with 256-wide
// panels (the planner's lists are far shorter; one that is not makes the block ineligible, get_plan() returns null).
size_t max_tasks(int n)
{
    const size_t tm = (size_t)n / TM;
    size_t nt = 0;
    for (size_t Mt = 1; Mt <= tm; ++Mt) nt += Mt * (Mt + 1) + Mt;
    return 4 * nt + 1024;
}

// Upper bound of the queue's workspace for any block of order <= n: with 256-wide panels everywhere step k
// (trailing order M tiles of 256) has M (M + 1) update tiles and M solves.
size_t ws_bytes(int n)
{
    if (!q_on()) return 0;
    const int nq = std::min(n, q_max()) / TM * TM;
    if (nq < q_min()) return 0;
    const size_t tm = (size_t)nq / TM, tn = (size_t)nq / TN;
    return 256 + Q_WORDS * 4 + pad256(tm * tn * 4 * VS) + pad256(tn * 4 * VS) + pad256((tm + 1) * 4) + pad256(max_tasks(nq) * 8) + 256;
}

Ws carve(void *base, int n)
{
    Ws w{};
    const size_t tm = (size_t)n / TM, tn = (size_t)n / TN;
    char *p = reinterpret_cast<char *>(((uintptr_t)base + 255) / 256 * 256);
    w.qs = reinterpret_cast<int *>(p);            p += Q_WORDS * 4;
    w.ver = reinterpret_cast<int *>(p);           p += pad256(tm * tn * 4 * VS);
    w.tver = reinterpret_cast<int *>(p);          p += pad256(tn * 4 * VS);
    w.zero_bytes = (size_t)(p - reinterpret_cast<char *>(w.qs));
    w.pstart = reinterpret_cast<int *>(p);        p += pad256((tm + 1) * 4);
    w.tasks = reinterpret_cast<unsigned *>(p);
    return w;
}

unsigned long long *trace_panel_base(int ntasks)
{
    return (g_trace && (size_t)ntasks <= g_trace_cap) ? g_trace + TRACE_STRIDE * g_trace_cap + 2 * TRACE_WORKERS : nullptr;
}

namespace {
Ws g_last_ws{};
int g_last_n = 0, g_last_ntasks = 0;
const Plan *g_last_plan = nullptr;
}
void remember(const Ws &w, int n, int ntasks) { g_last_ws = w; g_last_n = n; g_last_ntasks = ntasks; }
void remember_plan(const Plan *p) { g_last_plan = p; }

int postmortem(bool always)
{
    if (!g_last_n) return 0;
    const int n = g_last_n, tn = n / TN, tm = n / TM;
    int h64[64];
    SGPR_HIP(hipMemcpy(h64, g_last_ws.qs, sizeof(h64), hipMemcpyDeviceToHost));
    // the post-mortem words as they were numbered before the abort word got a cache line of its own:
    // [0] ticket head, [1] abort, [2..7] the worker that gave up, [8..12] the panel strip that gave up
    int h[32] = {};
    for (int q = 0; q < 8; ++q) h[q] = h64[q];
    h[1] = h64[Q_ABORT];
    for (int q = 8; q <= 12; ++q) h[q] = h64[Q_ABORT + q - 1];
    if (!h[1] && !always) return 0;
    fprintf(stderr, "cholq n=%d %s: head %d of %d tasks; worker: set %d task %08x (type %u k %u i %u j %u) ticket %d short-mask %x head-then %d | "
            "panel: set %d block %d idx0 %d cnt %d need %d\n", n, h[1] ? "GAVE UP" : "state", h[0], g_last_ntasks, h[2], (unsigned)h[4], (unsigned)h[4] >> 30,
            ((unsigned)h[4] >> 21) & 511u, ((unsigned)h[4] >> 11) & 1023u, (unsigned)h[4] & 2047u, h[6], (unsigned)h[5], h[7],
            h[8], h[9], h[10], h[11], h[12]);
    std::vector<int> tv((size_t)tn), vv((size_t)tm * tn);
    {
        std::vector<int> raw(std::max(tv.size(), vv.size()) * VS);
        SGPR_HIP(hipMemcpy(raw.data(), g_last_ws.tver, tv.size() * VS * 4, hipMemcpyDeviceToHost));
        for (size_t q = 0; q < tv.size(); ++q) tv[q] = raw[q * VS];
        SGPR_HIP(hipMemcpy(raw.data(), g_last_ws.ver, vv.size() * VS * 4, hipMemcpyDeviceToHost));
        for (size_t q = 0; q < vv.size(); ++q) vv[q] = raw[q * VS];
    }
    if (g_last_plan && g_last_plan->n == n) {
        // the first tasks of the list whose effect is not there, with the words they wait for as they stand now
        const Plan &pl = *g_last_plan;
        int shown = 0;
        for (size_t t = 0; 2 * t < pl.tasks.size() && shown < 6; ++t) {
            const unsigned w0 = pl.tasks[2 * t], w1 = pl.tasks[2 * t + 1];
            const int type = (int)(w0 >> 30), k = (int)((w0 >> 21) & 511u), i = (int)((w0 >> 11) & 1023u), j = (int)(w0 & 2047u);
            if (type == TASK_U) {
                const int a = (int)(w1 >> 16), b = (int)(w1 & 0xffffu);
                if (vv[(size_t)i * tn + j] >= b) continue;
                fprintf(stderr, "  undone: ticket %zu update (%d,%d) [%d,%d): ver %d, tver rows %d %d, tver column strip %d\n", t, i, j, a, b,
                        vv[(size_t)i * tn + j], tv[2 * i], tv[2 * i + 1], tv[j]);
            } else {
                const int fin = pl.starts[k + 1] / LEAF;
                if (tv[2 * i] >= fin) continue;
                fprintf(stderr, "  undone: ticket %zu solve panel %d row tile %d: tver %d %d (want %d); ver of its tiles:", t, k, i, tv[2 * i], tv[2 * i + 1], fin);
                for (int c = 0; c < (pl.starts[k + 1] - pl.starts[k]) / LEAF; ++c) fprintf(stderr, " %d", vv[(size_t)i * tn + pl.starts[k] / TN + c]);
                fprintf(stderr, " (want %d)\n", pl.starts[k] / LEAF);
            }
            ++shown;
        }
    }
    fprintf(stderr, "  tver:");
    for (size_t r = 0; r < tv.size() && r < 256; ++r) fprintf(stderr, " %d", tv[r]);
    fprintf(stderr, "\n  ver (row tiles x column tiles, mod 36):\n");
    for (int i = 0; i < tm && i < 48; ++i) {
        fprintf(stderr, "   ");
        for (int j = 0; j < tn && j < 96; ++j) fprintf(stderr, "%c", "0123456789abcdefghijklmnopqrstuvwxyz"[vv[(size_t)i * tn + j] % 36]);
        fprintf(stderr, "\n");
    }
    return h[1];
}
int *abort_word(const Ws &w) { return w.qs + Q_ABORT; }

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
    const unsigned *tasks;      // two words per task
    int ntasks;
    const int *pstart;
    int *qs, *ver, *tver;
    const int *flags;           // the panel kernel's hand-off words of this block: panel starting at leaf column t at flags + t * pstride
    int pstride;
    const double *inv;          // leaf inverses of this block
    int *dinfo;
    unsigned long long *trace;  // 4 words per ticket, or null
    unsigned long long *census; // 2 words per worker workgroup, or null
    unsigned pollcap;           // longest pause between two looks of a waiting workgroup, in units of ~3.4 us
    int last;                   // the last worker instance of the factorisation: it never drains, it waits (up to Q_WAIT_LIMIT)
};

constexpr int QT = 512;
constexpr unsigned long long Q_WAIT_LIMIT = 3ull * 100000000ull;      // 3 s of the 100 MHz real-time counter (round 3: 20 s).  Every earlier
                                                                       // instance has drained after 3 ms without progress; the last one waits this long for a
                                                                       // panel side that is kept off its CUs, then gives up -- and the caller factors again
                                                                       // with the look-ahead driver (capi.hip)
constexpr unsigned long long Q_GIVEUP_TICKS = 300000ull;               // 3 ms without any publish anywhere: drain this instance

// is the task behind a ticket already done?  (after a rewind the head passes over tasks that were finished out of order)
__device__ __forceinline__ bool task_done(const int *ver, const int *tver, int tn, const int *pstart, unsigned tk, unsigned tk1)
{
    const int type = (int)(tk >> 30), k = (int)((tk >> 21) & 511u), i = (int)((tk >> 11) & 1023u), j = (int)(tk & 2047u);
    if (type == TASK_U)
        return __hip_atomic_load((gint *)(ver + ((size_t)i * tn + j) * VS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (int)(tk1 & 0xffffu);
    return __hip_atomic_load((gint *)(tver + (size_t)(2 * i) * VS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= pstart[k + 1] / (int)LEAF;
}

// between two worker instances: the ticket head goes back to the first task that is not done, the drain word is cleared
__global__ __launch_bounds__(1024) void rewind_kernel(const QArgs a)
{
    __shared__ int first;
    if (threadIdx.x == 0) first = a.ntasks;
    __syncthreads();
    if (__hip_atomic_load((gint *)(a.qs + Q_DRAIN), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
        __hip_atomic_load((gint *)a.qs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.ntasks)
        return;                                   // the instance before ran the list to its end
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int t = threadIdx.x; t < a.ntasks; t += blockDim.x)
        if (!task_done(a.ver, a.tver, a.tn, a.pstart, a.tasks[2 * (size_t)t], a.tasks[2 * (size_t)t + 1])) { atomicMin(&first, t); break; }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store((gint *)a.qs, first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((gint *)(a.qs + Q_DRAIN), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(QT, 2) void chol_queue_kernel(const QArgs a)
{
    constexpr int LDA_S = TM + PAD, LDB_S = TN + PAD;
    __shared__ double smem[2 * BK * (LDA_S + LDB_S)];
    __shared__ int sh[4];
    const int tid = threadIdx.x;
    if (tid == 0) sh[0] = atomicAdd(a.qs, 1);
    if (a.census && tid == 0 && blockIdx.x < TRACE_WORKERS && a.census[2 * blockIdx.x + 1] == 0) {      // (the first instance's)
        a.census[2 * blockIdx.x] = ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) |
                                   (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        a.census[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    int t = sh[0];
    __syncthreads();                             // (sh[0] is written again below, by one lane, possibly before a slow wave has read it)
    // Tickets are worked in list order, and a workgroup WAITS for the task it has drawn.  Tried in round 4: parking a task whose
    // inputs are not there yet (up to three per workgroup, re-examined oldest first) and drawing on -- n = 16384: 27.7 ms in
    // strict order, 27.9 - 28.7 with one parked task after 30 - 300 us of patience, 32 - 33 with two, 45 without patience.  The
    // list order IS the priority order: a worker that helps itself to a long bulk update while the task next to the chain is
    // 20 us from being ready costs the chain 400.
    while (t < a.ntasks) {
        const unsigned tk = a.tasks[2 * (size_t)t], tk1 = a.tasks[2 * (size_t)t + 1];
        // this instance is being drained (Q_DRAIN), or the task was finished before a rewind: nothing to do for this ticket
        if (tid == 0) {
            const int dr = __hip_atomic_load((gint *)(a.qs + Q_DRAIN), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh[2] = dr ? 2 : (task_done(a.ver, a.tver, a.tn, a.pstart, tk, tk1) ? 1 : 0);
            if (sh[2] == 1) sh[0] = atomicAdd(a.qs, 1);
        }
        __syncthreads();
        const int skip = sh[2];
        const int tskip = sh[0];
        __syncthreads();
        if (skip == 2) break;
        if (skip == 1) { t = tskip; continue; }
        const int type = (int)(tk >> 30), k = (int)((tk >> 21) & 511u), i = (int)((tk >> 11) & 1023u), j = (int)(tk & 2047u);
        // update: leaf columns [ca, cb) of L; solve: panel k = columns [s0, s0 + w)
        const int ca = (int)(tk1 >> 16), cb = (int)(tk1 & 0xffffu);
        const int s0 = (type == TASK_T) ? a.pstart[k] : ca * LEAF;
        const int w = (type == TASK_T) ? a.pstart[k + 1] - s0 : (cb - ca) * LEAF;
        const int W = w / LEAF, j0 = s0 / TN;
        const bool tr = a.trace != nullptr && tid == 0;
        if (tr) a.trace[8 * (size_t)t] = __builtin_amdgcn_s_memrealtime();
        // the ticket after this one is drawn now and used at the bottom: its round trip hides under the task
        int tnext = 0;
        if (tid == 0) tnext = atomicAdd(a.qs, 1);
        // ---- inputs ready?  One lane per word, relaxed polls, then ONE agent-scope acquire for the workgroup
        if (tid < 64) {
            const int *p = nullptr;
            int need = 0;
            if (type == TASK_U) {
                if (tid == 0)      { p = a.ver + ((size_t)i * a.tn + j) * VS; need = ca; }     // the tile has taken columns [0, ca)
                else if (tid == 1) { p = a.tver + (size_t)(2 * i) * VS;     need = cb; }     // rows of L final through column cb
                else if (tid == 2) { p = a.tver + (size_t)(2 * i + 1) * VS; need = cb; }
                else if (tid == 3) { p = a.tver + (size_t)j * VS;           need = cb; }
            } else {
                if (tid < W)          { p = a.ver + ((size_t)i * a.tn + j0 + tid) * VS; need = s0 / LEAF; }
                else if (tid < 2 * W) { p = a.flags + (size_t)j0 * a.pstride + 2 + (tid - W); need = 1; }   // I[c]: leaf c inverted
            }
            unsigned spins = 0;
            int ok = 1;
            bool timed_out = false;
            const unsigned long long twait0 = __builtin_amdgcn_s_memrealtime();
            unsigned long long tprog = twait0;       // when the progress word last changed under this wait
            int prog = __hip_atomic_load((gint *)(a.qs + Q_PROG), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                const int v = p ? __hip_atomic_load((gint *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                if (__all(v >= need)) break;
                // Back off, but not far: the pause grows from 0.2 us to ~7 us (SGPR_Q_POLLCAP x 3.4 us).  Round 3 first let it
                // grow to 27 us, in the belief that hundreds of pollers starve the loads of the workgroups they wait for -- the
                // stalls that suggested it were workgroups that had been switched out; the longer pause cost 1 - 2 %.
                ++spins;
                const unsigned reps = spins < 5u ? 0u : (spins < 13u ? min(spins - 4u, a.pollcap) : a.pollcap);
                if (reps == 0u) __builtin_amdgcn_s_sleep(8 << 2);
                for (unsigned r = 0; r < reps; ++r) __builtin_amdgcn_s_sleep(127);
                if (spins < 5u) continue;
                // give up when somebody else has, or after Q_WAIT_LIMIT of REAL time (a bound counted in polls would depend on how the
                // polls are spaced, and a wave can stand still for a while without any fault of the program)
                const int ab = __hip_atomic_load((gint *)(a.qs + Q_ABORT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long tnow = __builtin_amdgcn_s_memrealtime();
                timed_out = tnow - twait0 > Q_WAIT_LIMIT;
                if (ab != 0 || timed_out) { ok = 0; break; }
                // nothing published anywhere for Q_GIVEUP_TICKS: somebody this grid waits for is not running.
                // Drain the instance: the waiters leave, whoever was stranded gets a CU, finishes and leaves too.
                const int pg = __hip_atomic_load((gint *)(a.qs + Q_PROG), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pg != prog) { prog = pg; tprog = tnow; }
                int dr = __hip_atomic_load((gint *)(a.qs + Q_DRAIN), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!dr && !a.last && tnow - tprog > Q_GIVEUP_TICKS) {
                    __hip_atomic_store((gint *)(a.qs + Q_DRAIN), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dr = 1;
                }
                if (dr) { ok = 2; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ok == 0 && timed_out) {
                // post-mortem (sgpr_probe_queue_postmortem): the first task that gave up, and which of its words were short
                const int v = p ? __hip_atomic_load((gint *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                const unsigned long long short_mask = __ballot(v < need);
                if (tid == 0 && atomicCAS(a.qs + 2, 0, 1) == 0) {
                    a.qs[4] = (int)tk; a.qs[5] = (int)(short_mask & 0xffffffffu); a.qs[6] = t; a.qs[7] = a.qs[0]; a.qs[3] = (int)tk1;
                }
            }
            if (tid == 0) {
                if (ok == 0) {
                    if (timed_out) atomicCAS(a.dinfo, 0, POTRF_HANDOFF_TIMEOUT);
                    __hip_atomic_store((gint *)(a.qs + Q_ABORT), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sh[1] = ok;
            }
        }
        __syncthreads();
        if (sh[1] != 1) break;                   // workgroup-uniform: given up (0), or this instance is draining (2: the ticket stays undone)
        if (tr) a.trace[8 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
        // ---- the task's products (one call site of the k-loop body)
        const int nprod = (type == TASK_U) ? 1 : 2 * W - 1;
        for (int p = 0; p < nprod; ++p) {
            GemmArgs g{};
            g.m = a.n; g.n = a.n; g.lda = a.lda; g.ldc = a.lda; g.stamps = nullptr;
            g.kdone = a.trace ? a.trace + 8 * (size_t)t + 4 : nullptr;
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
        if (tr) a.trace[8 * (size_t)t + 5] = __builtin_amdgcn_s_memrealtime();       // wave 0 is out of the products
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tr) a.trace[8 * (size_t)t + 6] = __builtin_amdgcn_s_memrealtime();       // every wave's stores have drained
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tr) a.trace[8 * (size_t)t + 7] = __builtin_amdgcn_s_memrealtime();   // the L2 write-back is through
            if (type == TASK_U) {
                __hip_atomic_store((gint *)(a.ver + ((size_t)i * a.tn + j) * VS), cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const int fin = (s0 + w) / LEAF;
                __hip_atomic_store((gint *)(a.tver + (size_t)(2 * i) * VS), fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store((gint *)(a.tver + (size_t)(2 * i + 1) * VS), fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_fetch_add((gint *)(a.qs + Q_PROG), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tr) {
                a.trace[8 * (size_t)t + 2] = __builtin_amdgcn_s_memrealtime();
                // (word 1 of a task is < 2^27: the XCC id rides in bits 60..63)
                a.trace[8 * (size_t)t + 3] = ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 60) |
                                             ((unsigned long long)tk1 << 32) | tk;
            }
            sh[0] = tnext;
        }
        __syncthreads();
        t = sh[0];
    }
}

}  // namespace

static std::atomic<int> g_force_giveup{0};
void force_giveup(int on) { g_force_giveup.store(on); }
bool forced_giveup() { return g_force_giveup.load() != 0; }

int prepare(const Plan &p, const Ws &w, hipStream_t st)
{
    SGPR_HIP(hipMemsetAsync(w.qs, 0, w.zero_bytes, st));
    // tests only (sgpr_probe_queue_force_giveup): the give-up word is raised before anybody runs -- every worker and panel
    // strip leaves at its first wait, the factorisation ends unfinished, exactly as after a hand-off that timed out
    if (g_force_giveup.load()) SGPR_HIP(hipMemsetAsync(w.qs + Q_ABORT, 1, sizeof(int), st));
    SGPR_HIP(hipMemcpyAsync(w.pstart, p.pinned, (size_t)(p.nblk + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    SGPR_HIP(hipMemcpyAsync(w.tasks, p.pinned + p.nblk + 1, p.tasks.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));   // two words per task
    return 0;
}

int launch_workers(const Plan &p, const Ws &w, double *A, size_t lda, const double *inv, const int *flags, int *dinfo,
                   int pflag_stride, hipStream_t st)
{
    if (p.tasks.empty()) return 0;
    QArgs a{};
    a.A = A; a.lda = lda; a.n = p.n; a.tn = p.n / TN;
    a.tasks = w.tasks; a.ntasks = (int)(p.tasks.size() / 2); a.pstart = w.pstart;
    a.qs = w.qs; a.ver = w.ver; a.tver = w.tver;
    a.flags = flags; a.pstride = pflag_stride; a.inv = inv; a.dinfo = dinfo;
    a.trace = (g_trace && p.tasks.size() / 2 <= g_trace_cap) ? g_trace : nullptr;
    a.census = a.trace ? g_trace + TRACE_STRIDE * g_trace_cap : nullptr;
    static const int pollcap = std::max(1, std::min(8, (int)tune("q_pollcap", 2)));
    a.pollcap = (unsigned)pollcap;
    const int grid = std::max(1, std::min(p.nworkers, (int)(p.tasks.size() / 2)));
    // Q_INSTANCES worker kernels back to back, a rewind of the ticket head between them: the first normally runs the whole
    // list and the others leave at once (~5 us each); when an instance drains (Q_DRAIN) the next one carries on
    const int instances = std::min(32, Q_INSTANCES + (int)(p.model_us / 8000.0));   // room for one drain per ~8 ms of the plan
    for (int inst = 0; inst < instances; ++inst) {
        if (inst) {
            hipLaunchKernelGGL(rewind_kernel, dim3(1), dim3(1024), 0, st, a);
            SGPR_CHECK_LAUNCH();
        }
        // (the last instance does not drain: if something keeps the panel side off its CUs for longer than all the
        // instances together -- another stream's long kernel, say -- it waits like the single instance of the first version did)
        a.last = inst == instances - 1 ? 1 : 0;
        hipLaunchKernelGGL(chol_queue_kernel, dim3(grid), dim3(QT), 0, st, a);
        SGPR_CHECK_LAUNCH();
    }
    return 0;
}

}  // namespace cholq
}  // namespace sgpr

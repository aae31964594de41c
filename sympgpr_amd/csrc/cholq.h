// cholq.h -- the task-queue form of the blocked Cholesky factorisation (mid-size orders).
//
// scipy.linalg.cholesky(Ky, lower=True) -> LAPACK dpotrf (python/functions/func.py:166,184,193 of the
// reference) for orders where neither the recursion nor a launch-per-step schedule keeps the chip busy:
// ONE persistent grid of workers pulls 256 x 128 tile tasks of the whole factorisation from an ordered list
// (trailing updates U and the panel's rows-below solves T), each task waiting on device-side version counters
// instead of kernel boundaries, while the chain of diagonal blocks runs beside it in the persistent panel
// kernel (chol.hip) on a handful of CUs the worker grid leaves free.  No per-step fill / drain, no tile
// round-up per launch, and the panel stream is never starved of CUs.
//
// This header is shared by cholq.hip (plan builder, worker kernel) and chol.hip (the driver that launches the
// panel kernels beside the workers).
#pragma once
#include <vector>

#include "common.h"

namespace sgpr {
namespace cholq {

constexpr int TM = 256, TN = 128;          // the workers' tile of the trailing matrix
enum { TASK_U = 0, TASK_T = 1 };

// One task = two words.  Word 0: [31:30] type, [29:21] panel k (solves), [20:11] row tile i (256 rows), [10:0] column
// tile j (128 columns); word 1 (updates): [31:16] a, [15:0] b in units of 128 columns.
//   U(i, j, a, b): C(i, j) -= L(i, a:b) L(j, a:b)^T     needs ver[i][j] >= a, tver[2i], tver[2i+1], tver[j] >= b; leaves ver[i][j] = b
//   T(k, i)      : A(i, panel k) := A(i, panel k) L_kk^-T   needs ver[i][columns of panel k] >= start of panel k, the
//                  panel's leaves inverted; leaves tver[2i] = tver[2i+1] = end of panel k
// ver[i][j]: leading 128-column blocks of L already applied to tile (i, j); tver[r]: leading blocks of row strip r (128
// rows) that are final.  The rows of the NEXT diagonal block have no T task: the panel kernel solves them beside its
// chain and sets their tver itself.
inline unsigned pack(int type, int k, int i, int j) { return ((unsigned)type << 30) | ((unsigned)k << 21) | ((unsigned)i << 11) | (unsigned)j; }
inline int task_type(unsigned t) { return (int)(t >> 30); }
inline int task_k(unsigned t) { return (int)((t >> 21) & 511u); }
inline int task_i(unsigned t) { return (int)((t >> 11) & 1023u); }
inline int task_j(unsigned t) { return (int)(t & 2047u); }
constexpr int MAX_ORDER = 131072;          // 9-bit panel index at >= 256 columns per panel, 10-bit row tile

struct Plan {
    int n = 0, nblk = 0, wmax = 0, nworkers = 0;
    int nq = 0;                            // panels 0 .. nq-1 are factored by the queue; the block that is left (order n - starts[nq],
                                           // every update of the first nq panels applied) goes to the look-ahead driver
    std::vector<int> starts;               // nblk + 1 panel boundaries (multiples of 256)
    std::vector<unsigned> tasks;           // two words per task, in ticket order: every input of a task comes from a smaller ticket
    double flop = 0.0;                     // algorithmic flop of the tasks (2k per updated element on / below the diagonal)
    double model_us = 0.0;                 // the planner's own estimate of the factorisation time
    unsigned *pinned = nullptr;            // page-locked copy for the upload: [starts | tasks]
};

// panel boundaries of the default schedule for order n (tunable "q_w": sgpr_probe_tune)
std::vector<int> default_starts(int n);
// host only: the ordered task list for given panel boundaries (all multiples of 256, starts[0] = 0, back() = n)
// nq < 0: all panels
int build_plan(int n, const std::vector<int> &starts, int nworkers, Plan &out, int nq = -1);
// the hand-over point of the schedule: with the tunable "q_tail" = <rows> (default 0: the queue runs everything; sgpr_probe_tune) the queue runs the
// panels while more than that many rows are left and the look-ahead driver factors the rest.  Measured: no gain
// (n = 16384: 30.6 - 31.6 ms against 30.0 for the whole factorisation in the queue and 32.2 for the look-ahead driver), and
// the host has to wait between the two parts.
int default_nq(int n, const std::vector<int> &starts);
// cached per (n, nworkers); nullptr on failure (error text set)
const Plan *get_plan(int n, int nworkers);

// bytes of the queue's part of the factor workspace for order n (0: the queue form is not used for this order)
size_t ws_bytes(int n);
size_t max_tasks(int n);
bool eligible(int n);

// The ticket head is hit by a returning atomic from every workgroup at every task; the give-up word is READ by every
// waiting workgroup: a line each.  (Round 3 took the rare stalls of this driver for memory contention for a long time and
// spread its state words over cache lines because of that; they were workgroups that had been switched out, DESIGN 3.9.
// The layout stays: it costs nothing.)
constexpr int Q_ABORT = 32;
// Third and fourth line of the state words.  Q_PROG: bumped at every publish (workers and panel strips) -- "something moved".
// Q_DRAIN: set by a worker that has seen nothing move for Q_GIVEUP_TICKS; every worker that is NOT inside a task's
// products then leaves (its ticket is simply not done), the ones that are finish their task and leave, the kernel instance
// ends, and the next instance (already enqueued behind a rewind of the ticket head) carries on.  Why: DESIGN 3.9 -- when the
// platform has switched the queues out and in, a few workgroups may not get their CU back until somebody else leaves one.
constexpr int Q_PROG = 64, Q_DRAIN = 96, Q_WORDS = 128;
constexpr int Q_INSTANCES = 6;             // worker kernel instances enqueued per factorisation at least (+1 per ~8 ms of plan; all but the first normally find nothing to do)
// ... and every version word too: VS ints apart = one 128-byte line per word.  Packed, the 128 words that say how far
// each row strip is solved sat on four lines that every waiting workgroup of the chip polled.
constexpr size_t VS = 32;                // qs[32..63]: give-up word + the panel kernel's post-mortem words

struct Ws {                                // pointers into the queue's part of the workspace
    int *qs;                               // Q_WORDS ints: [0] ticket head, [2..7] post-mortem of the worker that gave up, [Q_ABORT] give-up word, [Q_PROG], [Q_DRAIN]
    int *ver;                              // tm x tn words (VS ints apart): leading 128-column blocks of L applied to tile (i, j)
    int *tver;                             // tn words (VS ints apart): leading blocks final per 128-row strip
    int *pstart;                           // nblk + 1
    unsigned *tasks;
    size_t zero_bytes;                     // qs .. tver: cleared before every factorisation
};
Ws carve(void *base, int n);

// enqueue upload + clear on `st`; then the worker grid (call after the first panel kernel is on its stream)
int prepare(const Plan &p, const Ws &w, hipStream_t st);
int launch_workers(const Plan &p, const Ws &w, double *A, size_t lda, const double *inv, const int *flags, int *dinfo,
                   int pflag_stride, hipStream_t st);

// diagnostics: the last queue workspace used by this process, and a dump of its state words (head, abort, the
// first task / panel strip that gave up, tver, the ver map) to stderr; `always` = also when nothing gave up
void remember(const Ws &w, int n, int ntasks);
void remember_plan(const Plan *p);
int postmortem(bool always);
int *abort_word(const Ws &w);
bool forced_giveup();                      // tests only: see force_giveup
void force_giveup(int on);                 // tests only (libsympgpr_probe.so): the next queue factorisations give up at once

// diagnostics (libsympgpr_probe.so): per-task time stamps of the next factorisation(s) in this process
void set_trace(unsigned long long *dev_buf, size_t capacity_tasks);
// layout of the trace buffer: 8 words per ticket (free, inputs there, published, task words, ticket returned, out of the
// products, stores drained, write-back through), then 2 per worker workgroup (place, start), then 4 per workgroup of every
// panel (place, start, end, strip), 32 workgroups per panel
constexpr size_t TRACE_WORKERS = 1024, TRACE_PANELS = 512, TRACE_PANEL_WGS = 32;
constexpr size_t TRACE_STRIDE = 8;
inline size_t trace_words(size_t cap) { return TRACE_STRIDE * cap + 2 * TRACE_WORKERS + 4 * TRACE_PANELS * TRACE_PANEL_WGS; }
unsigned long long *trace_panel_base(int ntasks);   // null when no trace is being taken

}  // namespace cholq
}  // namespace sgpr

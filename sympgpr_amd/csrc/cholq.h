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

// One task, packed: [31:30] type, [29:21] panel k, [20:11] row tile i (256 rows), [10:0] column tile j (128 columns)
//   U(k, i, j): C(i, j) -= L(i, panel k) L(j, panel k)^T        needs tver[2i], tver[2i+1], tver[j] > k, ver[i][j] == k
//   T(k, i)   : A(i, panel k) := A(i, panel k) L_kk^-T          needs ver[i][columns of panel k] == k, the panel's leaves
// (ver: updates applied per tile; tver: panels solved per 128-row strip.  The rows of the NEXT diagonal block have no
// T task: the panel kernel solves them beside its chain and bumps their tver itself.)
inline unsigned pack(int type, int k, int i, int j) { return ((unsigned)type << 30) | ((unsigned)k << 21) | ((unsigned)i << 11) | (unsigned)j; }
inline int task_type(unsigned t) { return (int)(t >> 30); }
inline int task_k(unsigned t) { return (int)((t >> 21) & 511u); }
inline int task_i(unsigned t) { return (int)((t >> 11) & 1023u); }
inline int task_j(unsigned t) { return (int)(t & 2047u); }
constexpr int MAX_ORDER = 131072;          // 9-bit panel index at >= 256 columns per panel, 10-bit row tile

struct Plan {
    int n = 0, nblk = 0, wmax = 0, nworkers = 0;
    std::vector<int> starts;               // nblk + 1 panel boundaries (multiples of 256)
    std::vector<unsigned> tasks;           // in ticket order: every dependency of a task has a smaller ticket
    double flop = 0.0;                     // algorithmic flop of the tasks (2k per updated element on / below the diagonal)
    unsigned *pinned = nullptr;            // page-locked copy for the upload: [starts | tasks]
};

// panel boundaries of the default schedule for order n (SGPR_Q_* override)
std::vector<int> default_starts(int n);
// host only: the ordered task list for given panel boundaries (all multiples of 256, starts[0] = 0, back() = n)
int build_plan(int n, const std::vector<int> &starts, int nworkers, Plan &out);
// cached per (n, nworkers); nullptr on failure (error text set)
const Plan *get_plan(int n, int nworkers);

// bytes of the queue's part of the factor workspace for order n (0: the queue form is not used for this order)
size_t ws_bytes(int n);
bool eligible(int n);

struct Ws {                                // pointers into the queue's part of the workspace
    int *qs;                               // [0] ticket head, [1] abort, [2..] spare
    int *ver;                              // tm x tn update counts per tile
    int *tver;                             // tn: panels solved per 128-row strip
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
int postmortem(bool always);

// diagnostics (libsympgpr_probe.so): per-task time stamps of the next factorisation(s) in this process
void set_trace(unsigned long long *dev_buf, size_t capacity_tasks);
// layout of the trace buffer: 4 words per ticket, then 2 per worker workgroup (place, start), then 4 per workgroup of
// every panel kernel (place, start, end, strip), 32 workgroups per panel
constexpr size_t TRACE_WORKERS = 1024, TRACE_PANELS = 512, TRACE_PANEL_WGS = 32;
inline size_t trace_words(size_t cap) { return 4 * cap + 2 * TRACE_WORKERS + 4 * TRACE_PANELS * TRACE_PANEL_WGS; }
unsigned long long *trace_panel_base(int ntasks);   // null when no trace is being taken

}  // namespace cholq
}  // namespace sgpr

/*
 * sympgpr_probe.h -- C ABI of libsympgpr_probe.so: roofline calibration probes and kernel
 * diagnostics for the MI355X build of SympGPR's training core.  MEASUREMENT AIDS ONLY: nothing
 * here replaces an interface of the reference, a maintainer binding include/sympgpr_hip.h never
 * loads this library, and no product path calls it (tools/ and bench.py's calibration do).
 * Same conventions as sympgpr_hip.h (0 = ok, < 0 = SGPR_E_*).
 */
#ifndef SYMPGPR_PROBE_H
#define SYMPGPR_PROBE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* a register-only fp64 MFMA issue loop with
 * `waves_per_simd` waves on every SIMD, and a streaming 16-B/lane write of `bytes` bytes. */
int sgpr_probe_mfma_f64(int waves_per_simd, int iters, double *tflops);
int sgpr_probe_hbm_write(size_t bytes, int reps, double *gbs);
/* out3: TFLOP/s, shader cycles per MFMA per SIMD, shader clock (GHz) held during the loop */
int sgpr_probe_mfma_clock(int nacc, int waves_per_simd, int iters, double *out3);
/* one synthetic C -= A B^T with per-workgroup stamps: TFLOP/s, median k-loop cycles per
 * workgroup, median shader clock (GHz), k-steps per workgroup */
int sgpr_probe_gemm(int m, int n, int k, int lower, double *out4);
/* switches for the following sgpr_probe_gemm calls OF THIS THREAD (nothing in the product library
 * changes): 8 = 128x128 tile shape, 16 = register-staged body; 0 = normal */
int sgpr_probe_gemm_debug(int bits);
/* shader cycles per phase of one 128x128 leaf factorisation: load, diag block, panel rows,
 * trailing update, write-back, inverse diag, inverse rows, final store */
int sgpr_probe_leaf(double *out8);
/* cycles per step of eight dependent micro-sequences on one wave (the building blocks of the leaf's diagonal-block step) */
int sgpr_probe_lat(double *out8);
/* HW_REG_XCC_ID of each workgroup of a 1-D grid of 512-thread blocks (checks the tile map's `id % 8`) */
int sgpr_probe_xcc(int nblocks, int *host_out);
/* the same on a stream restricted by a CU mask (hipExtStreamCreateWithCUMask): out[2b] = XCC id, out[2b+1] = HW_ID */
int sgpr_probe_cumask(const unsigned *mask_words, int nwords, int nblocks, int *host_out);

/* The OUTPUT OF THE CODE GENERATOR (tools/gen_kernels.py -> csrc/generated/pair_generated.h), evaluated as
 * it stands: out[i] = f(xa[i], ya[i], xb[i], yb[i], l...) without the sig factor; `which` as in
 * sgpr_kernel_eval_host (0..3 = k, d2k/dxdx0, d2k/dydy0, d2k/dxdy0; | 4: d/dlx; | 8: d/dly).  The product
 * library's hand-optimised kernels are diffed against this in tests/test_gpu_generated.py. */
int sgpr_probe_generated_eval(int family, int which, int m, const double *xa, const double *ya,
                              const double *xb, const double *yb, const double *l, int nl, double *out);

/* The task-queue Cholesky (csrc/cholq.h).  Host only: the ordered task list the worker grid would run for order n
 * with `nworkers` workers: counts[0] = panels, counts[1] = tasks, counts[2] = the planner's estimate of the time in
 * us; starts_out (panels + 1 boundaries) and tasks_out (TWO words per task, cholq.h: word 0 = [31:30] type 0 = update
 * / 1 = rows-below solve, [29:21] panel, [20:11] row tile of 256, [10:0] column tile of 128; word 1 = [31:16] first,
 * [15:0] one-past-last 128-column block of L an update applies) are filled up to the given capacities (max_tasks in
 * tasks).  tests/test_queue_plan.py replays the list on the CPU. */
int sgpr_probe_queue_plan(int n, int nworkers, int *starts_out, int max_starts, unsigned *tasks_out, int max_tasks,
                          int *counts);
/* ... with the hand-over point: the queue factors panels 0 .. nq-1 only and leaves the block behind them (all their updates
 * applied) to the look-ahead driver; nq < 0: the default of this order (SGPR_Q_TAIL).  counts has FOUR entries here:
 * counts[3] = nq used. */
int sgpr_probe_queue_plan_partial(int n, int nworkers, int nq, int *starts_out, int max_starts, unsigned *tasks_out,
                                  int max_tasks, int *counts);
/* per-task time stamps of the queue factorisations that follow in this process (8 words per ticket: 100 MHz real
 * time at ticket drawn / inputs ready / published, then task word 1 << 32 | task word 0); _end copies them out and
 * switches the recording off again.  (words 4..7: ticket of the next task returned, out of the products, stores drained, write-back through.)  Behind the 8 * max_tasks ticket words: 2 words per worker workgroup (1024: place
 * = XCC id << 32 | HW_ID, start) and 4 per workgroup of every panel kernel (512 panels x 32: place, start, end, strip);
 * `out` holds 8 * max_tasks + 2048 + 65536 words.  Returns the capacity / the number of words copied. */
int sgpr_probe_queue_trace_begin(int max_tasks);
int sgpr_probe_queue_trace_end(unsigned long long *out, int max_tasks);
int sgpr_probe_queue_trace_clear(void);   /* zero the stamps between two factorisations */
/* state words of the last task-queue factorisation of this process on stderr (ticket head, abort word, the first
 * task / panel strip that gave up waiting, version counters); returns the abort word */
int sgpr_probe_queue_postmortem(int always);
/* TESTS ONLY: while on, every task-queue factorisation of this process gives up before it starts (the give-up word is raised and
 * the factor's info word set as by a hand-off that timed out) -- to exercise the callers' retry with the other driver */
int sgpr_probe_queue_force_giveup(int on);
/* Experiment knobs of the product library (they were SGPR_* environment variables up to round 3): set BEFORE the code that reads
 * them runs for the first time in the process; each is read once.  Task-queue Cholesky: q_w (panel width, 512), q_tail (rows left
 * to the look-ahead driver, 0), q_kcap, q_leaf_us / q_pair_us / q_fixed_us / q_band0_us (planner's cost model), q_pollcap (2),
 * q_slack (CUs left empty, 0), q_nosync, q_debug.  Look-ahead driver: la_panel (3 = persistent panel kernel, 1 / 2 / 0 the
 * multi-launch panels), la_t0 / la_t1 / la_t2 (width thresholds).  GEMM: gemm_small_tile, gemm_small_mb.  Batched fits:
 * batch_two_min (512), batch_below (workgroups for the strips below a panel's diagonal block, 0 = one per strip; read per call).
 * Block solves: trsm_chain (workgroups of the chain class, 0 = built-in), trsm_piece (tiles per stream ticket, 0 = built-in: 16 up
 * to 128 strips, 128 above; read per call). */
int sgpr_probe_tune(const char *name, double value);
/* the last sgpr_applymap_host of this process: K*-row evaluations (residuals of the implicit equation + q updates) summed over
 * its orbits, and the number of workgroups that share one orbit for ntest orbits on n0 training points */
unsigned sgpr_probe_map_calls(void);
int sgpr_probe_map_team(int ntest, int n0);

/* the block solve's stream tickets as the host computes them (csrc/trsm.hip: piece_of): ticket of a solve with `strips` strips ->
 * {strip, piece, pieces of the strip, slot of the strip's first partial sum, 0 or the tile f whose fold the ticket is} for pieces
 * of at most `cap` tiles (strip = strips: past the end); and for `strips` strips {tickets, partial sums} */
int sgpr_probe_trsm_piece(int ticket, int cap, int strips, int out[5]);
int sgpr_probe_trsm_counts(int strips, int cap, unsigned long long out[2]);

/* co-residency census of two concurrent kernels (A: na workgroups of threads_a threads with lds_a bytes of LDS spinning
 * spin_a us on one stream, B likewise on a second, high-priority stream; optional CU masks): per workgroup XCC id,
 * HW_ID, start and end in 100 MHz ticks.  Answers "how many CUs must a persistent grid leave free, and where". */
int sgpr_probe_census(int na, int threads_a, int lds_a, int spin_a, const unsigned *mask_a, int nwords_a,
                      int nb, int threads_b, int lds_b, int spin_b, const unsigned *mask_b, int nwords_b,
                      unsigned long long *host_out);

#ifdef __cplusplus
}
#endif
#endif

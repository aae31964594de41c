"""Many task-queue factorisations back to back, post-mortem on the first failure: python tools/queue_stress.py [--trace] N reps [N reps ...]"""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import _lib as L
from sympgpr_amd.fit import SympFit
from bench import synth
probe = L.load_probe_library()
trace = "--trace" in sys.argv
args = [int(a) for a in sys.argv[1:] if not a.startswith("--")]
cap = 1 << 17
nwords = 8 * cap + 2048 + 65536
if trace:
    buf = (C.c_ulonglong * nwords)()
    assert probe.sgpr_probe_queue_trace_begin(cap) == cap
for N, reps in zip(args[0::2], args[1::2]):
    q, P, z, hyp, s2 = synth(N)
    with SympFit("A", q, P, z, hyp, s2, lower_only=False) as f:
        t0 = time.time()
        ts = []
        for r in range(reps):
            try:
                if trace:
                    probe.sgpr_probe_queue_trace_clear()
                f.build(); f.factor()
                ts.append(f.stage_ms()[1])
            except Exception as e:
                print("n=%d rep %d FAILED: %s" % (2 * N, r, e), flush=True)
                probe.sgpr_probe_queue_postmortem(1)
                if trace:
                    probe.sgpr_probe_queue_trace_end(buf, cap)
                    allw = np.frombuffer(buf, dtype=np.uint64, count=nwords)
                    tr = allw[:8 * cap].reshape(-1, 8)
                    wc = allw[8 * cap:8 * cap + 2048].reshape(-1, 2)
                    t0_ = int(wc[wc[:, 1] != 0][:, 1].min())
                    nz = np.where(tr[:, 0] != 0)[0]
                    print("trace: %d tickets reached their loop top, %d got their inputs, %d published; highest ticket seen %d" % (
                        len(nz), int((tr[:, 1] != 0).sum()), int((tr[:, 2] != 0).sum()), int(nz.max())))
                    started_not_ready = np.where((tr[:, 0] != 0) & (tr[:, 1] == 0))[0]
                    print("tickets waiting for inputs at the end (drawn us):", [(int(t), (int(tr[t, 0]) - t0_) / 100.0) for t in started_not_ready[:40]])
                    ready_not_done = np.where((tr[:, 1] != 0) & (tr[:, 2] == 0))[0]
                    print("tickets running at the end:", [(int(t), (int(tr[t, 1]) - t0_) / 100.0) for t in ready_not_done[:40]])
                    # for the tickets that wait: who produces the word they wait for, and when was it published?
                    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
                    import queue_sim as qs
                    starts, tasks, _ = qs.fetch_plan(2 * N, int((wc[:, 1] != 0).sum()))
                    info = [qs.unpack(t) for t in tasks]
                    for tkt in started_not_ready[:12]:
                        typ, k, i, j, a, b = info[tkt]
                        if typ != 0:
                            print("  ticket %d: solve panel %d row tile %d, drawn %.0f us" % (tkt, k, i, (int(tr[tkt, 0]) - t0_) / 100.0))
                            continue
                        prod = [q for q in range(tkt) if info[q][0] == 0 and info[q][2] == i and info[q][3] == j and info[q][5] == a]
                        ts_ = [(q, (int(tr[q, 0]) - t0_) / 100.0, (int(tr[q, 1]) - t0_) / 100.0 if tr[q, 1] else -1, (int(tr[q, 2]) - t0_) / 100.0 if tr[q, 2] else -1) for q in prod]
                        # producers of the tver words: solves of row tile i / j // 2 ending at >= b
                        print("  ticket %d: update (%d,%d) [%d,%d) drawn %.0f us; producer of ver=%d (ticket, drawn, start, published): %s" % (
                            tkt, i, j, a, b, (int(tr[tkt, 0]) - t0_) / 100.0, a, ts_))
                    # where did the tasks that took longest run?
                    dur = (tr[:, 2].astype(np.int64) - tr[:, 1].astype(np.int64)) / 100.0
                    slow = np.where((tr[:, 2] != 0) & (dur > 100000))[0]
                    xs = (tr[slow, 3] >> 60).astype(int)
                    print("tasks that took > 0.1 s: %d, on XCCs %s (count per XCC); their start times span %.0f..%.0f us" % (
                        len(slow), np.bincount(xs, minlength=8).tolist(), (tr[slow, 1].astype(np.int64).min() - t0_) / 100.0 if len(slow) else 0,
                        (tr[slow, 1].astype(np.int64).max() - t0_) / 100.0 if len(slow) else 0))
                    for q in slow[:20]:
                        st = [(int(tr[q, c_]) - t0_) / 100.0 if tr[q, c_] else -1 for c_ in (0, 1, 4, 5, 6, 7, 2)]
                        print("   slow ticket %d on XCC %d: free %.0f  inputs %.0f  left the k-loop %.0f  out of products %.0f  drained %.0f  written back %.0f  published %.0f" % ((q, int(tr[q, 3] >> 60)) + tuple(st)))
                    fast = np.where((tr[:, 2] != 0) & (tr[:, 2].astype(np.int64) - t0_ > 3000000) & (dur < 100000))[0]
                    last = tr[:, 2].max()
                    print("last publish at %.0f us; workers started %d" % ((int(last) - t0_) / 100.0, int((wc[:, 1] != 0).sum())))
                    never = [t for t in range(int(nz.max())) if tr[t, 0] == 0]
                    print("tickets below the highest that never reached a loop top:", never[:40])
                sys.exit(1)
        f.solve(); a = f.alpha()
        op, oq = f.predict_rows(q[:256], P[:256])
        res = np.concatenate([op + s2 * a[:256] - z[:256], oq + s2 * a[N:N + 256] - z[N:N + 256]])
        print("n=%d: %d factorisations ok, factor min %.2f median %.2f max %.2f ms, resid %.1e (%.1f s)" % (
            2 * N, reps, min(ts), np.median(ts), max(ts), np.linalg.norm(res) / np.linalg.norm(z[:512]), time.time() - t0), flush=True)

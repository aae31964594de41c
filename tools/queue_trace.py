"""Per-task time stamps of one task-queue factorisation: python tools/queue_trace.py N_points [outfile.npz] [--first]
Prints the workers' utilisation, the mean task times by kind and panel, and where / when the workgroups of the
worker grid and of every panel kernel ran.  --first: trace the very first factorisation of the process."""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import _lib as L
from sympgpr_amd.fit import SympFit
from bench import synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
first = "--first" in sys.argv
N = int(args[0])
n = 2 * N
probe = L.load_probe_library()
q, P, z, hyp, s2 = synth(N)
cap = 1 << 19
nwords = 4 * cap + 2048 + 65536
buf = (C.c_ulonglong * nwords)()
failed = None
with SympFit("A", q, P, z, hyp, s2, lower_only=False) as f:
    ts = []
    if not first:
        f.run()
        for _ in range(3):
            f.build(); f.factor(); ts.append(f.stage_ms()[1])
    assert probe.sgpr_probe_queue_trace_begin(cap) == cap
    try:
        f.build(); f.factor(); traced_ms = f.stage_ms()[1]
    except Exception as e:
        failed = e
        traced_ms = float("nan")
        probe.sgpr_probe_queue_postmortem(1)
    nw = probe.sgpr_probe_queue_trace_end(buf, cap)
allw = np.frombuffer(buf, dtype=np.uint64, count=nwords)
tr = allw[:4 * cap].reshape(-1, 4)
wc = allw[4 * cap:4 * cap + 2048].reshape(-1, 2)
pc = allw[4 * cap + 2048:].reshape(512, 32, 4)
tr = tr[tr[:, 2] != 0]
print("n=%d factor %s ms (traced run %.2f ms), %d traced tasks%s" % (n, ("%.2f" % min(ts)) if ts else "-", traced_ms, len(tr),
                                                                     "  FAILED: %s" % failed if failed else ""))
wc = wc[wc[:, 1] != 0]
t0 = wc[:, 1].min() if len(wc) else (tr[:, 0].min() if len(tr) else 0)
def place(x):
    x = x.astype(np.int64)
    return (x >> 32) & 0xF, (x >> 13) & 7, (x >> 8) & 0xF      # xcc, se, cu
if len(wc):
    xcc, se, cu = place(wc[:, 0])
    cus = set(zip(xcc.tolist(), se.tolist(), cu.tolist()))
    print("workers: %d workgroups on %d distinct CUs, per XCC %s, started within %.1f us" % (
        len(wc), len(cus), np.bincount(xcc, minlength=8).tolist(), (wc[:, 1].max() - t0) / 100.0))
for k in range(512):
    m = pc[k][:, 1] != 0
    if not m.any():
        continue
    x = pc[k][m]
    xcc, se, cu = place(x[:, 0])
    shared = sum(1 for p_ in zip(xcc.tolist(), se.tolist(), cu.tolist()) if len(wc) and p_ in cus)
    st = (x[:, 1].astype(np.int64) - int(t0)) / 100.0
    en = (x[:, 2].astype(np.int64) - int(t0)) / 100.0
    order = np.argsort(x[:, 3])
    if k < 6 or failed:
        print("  panel %2d: %2d workgroups, start %8.0f..%8.0f us, end %8.0f..%8.0f us, on worker CUs: %d; strips/start/end: %s" % (
            k, m.sum(), st.min(), st.max(), en[en > -1e9].min() if (x[:, 2] != 0).any() else -1, en.max(), shared,
            " ".join("%d:%.0f-%.0f" % (int(x[o, 3]), st[o], en[o] if x[o, 2] else -1) for o in order)))
if len(tr):
    drawn = (tr[:, 0].astype(np.int64) - int(t0)) / 100.0
    ready = (tr[:, 1].astype(np.int64) - int(t0)) / 100.0
    done = (tr[:, 2].astype(np.int64) - int(t0)) / 100.0
    task = (tr[:, 3] & 0xFFFFFFFF).astype(np.uint32)
    typ = (task >> 30).astype(int); k = ((task >> 21) & 511).astype(int)
    print("worker kernel span %.2f ms; sum busy (ready->done) %.1f ms, sum wait (drawn->ready) %.1f ms" % (
        done.max() / 1e3, (done - ready).sum() / 1e3, (ready - drawn).sum() / 1e3))
    for kk in sorted(set(k)):
        for ty, name in ((1, "T"), (0, "U")):
            m = (k == kk) & (typ == ty)
            if m.any():
                print("  panel %2d %s: n=%5d  drawn %8.0f..%8.0f us  wait mean %6.1f max %7.1f  run mean %6.1f  (min %6.1f max %6.1f)" % (
                    kk, name, m.sum(), drawn[m].min(), drawn[m].max(), (ready - drawn)[m].mean(), (ready - drawn)[m].max(),
                    (done - ready)[m].mean(), (done - ready)[m].min(), (done - ready)[m].max()))
if len(args) > 1:
    np.savez_compressed(args[1], tr=tr, wc=wc, pc=pc[:64])

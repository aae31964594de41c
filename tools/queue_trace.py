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

for kv in [a for a in sys.argv[1:] if "=" in a]:      # name=value: experiment knobs (libsympgpr_probe.so), before anything runs
    L.check(L.load_probe_library().sgpr_probe_tune(kv.split("=")[0].encode(), float(kv.split("=")[1])))
sys.argv = [a for a in sys.argv if "=" not in a]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
first = "--first" in sys.argv
N = int(args[0])
n = 2 * N
probe = L.load_probe_library()
q, P, z, hyp, s2 = synth(N)
cap = 1 << 19
nwords = 8 * cap + 2048 + 65536
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
tr = allw[:8 * cap].reshape(-1, 8)
wc = allw[8 * cap:8 * cap + 2048].reshape(-1, 2)
pc = allw[8 * cap + 2048:].reshape(512, 32, 4)
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
    if k % 4 == 0 or failed:
        nd = int((x[:, 3] < 4).sum())
        print("  panel %2d: start %8.0f us; diagonal strips end %s; rows of the next block end %8.0f; on worker CUs: %d" % (
            k, st.min(), " ".join("%.0f" % en[o] for o in order[:nd]), en[order[nd:]].max() if len(order) > nd else -1, shared))
if len(tr):
    drawn = (tr[:, 0].astype(np.int64) - int(t0)) / 100.0
    ready = (tr[:, 1].astype(np.int64) - int(t0)) / 100.0
    done = (tr[:, 2].astype(np.int64) - int(t0)) / 100.0
    w0 = (tr[:, 3] & 0xFFFFFFFF).astype(np.uint32); w1 = ((tr[:, 3] >> 32) & 0x0FFFFFFF).astype(np.uint32)
    typ = (w0 >> 30).astype(int); kk = (w1 & 0xFFFF).astype(int) - (w1 >> 16).astype(int)
    run, wait = done - ready, ready - drawn
    print("worker kernel span %.2f ms; sum busy (start->done) %.1f ms, sum idle (free->start) %.1f ms" % (
        done.max() / 1e3, run.sum() / 1e3, wait.sum() / 1e3))
    for k_ in sorted(set(kk[typ == 0])):
        m = (typ == 0) & (kk == k_)
        print("  update k=%4d: n=%5d  run mean %6.1f (min %6.1f)  = k-steps + %5.1f   idle before: mean %6.1f" % (
            128 * k_, m.sum(), run[m].mean(), run[m].min(), run[m].mean() - 3.56 * 8 * k_, wait[m].mean()))
    for c in sorted(set(w1[typ == 1].tolist())):
        m = (typ == 1) & (w1 == c)
        print("  solve column %d: n=%5d  run mean %6.1f (min %6.1f)  idle before: mean %6.1f" % (c, m.sum(), run[m].mean(), run[m].min(), wait[m].mean()))
    edges = np.arange(0, done.max() + 500, 1000.0)
    nwk = max(len(wc), 1)
    print("busy fraction per ms:", " ".join("%.2f" % (np.clip(np.minimum(done, e1) - np.maximum(ready, e0), 0, None).sum() / (nwk * (e1 - e0)))
                                            for e0, e1 in zip(edges[:-1], edges[1:])))
if len(args) > 1:
    np.savez_compressed(args[1], tr=tr, wc=wc, pc=pc[:64])

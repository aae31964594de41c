"""Reproduce the intermittent give-up with the census on: python tools/queue_fail_hunt.py N [N ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import _lib as L
from sympgpr_amd.fit import SympFit
from bench import synth
probe = L.load_probe_library()
cap = 1 << 16
nwords = 8 * cap + 2048 + 65536
buf = (C.c_ulonglong * nwords)()
assert probe.sgpr_probe_queue_trace_begin(cap) == cap
failed = None
for N in [int(a) for a in sys.argv[1:]]:
    q, P, z, hyp, s2 = synth(N)
    try:
        with SympFit("A", q, P, z, hyp, s2, lower_only=False) as f:
            f.run()
            print("n=%d ok" % (2 * N), flush=True)
    except Exception as e:
        failed = (N, e)
        print("n=%d FAILED %s" % (2 * N, e), flush=True)
        probe.sgpr_probe_queue_postmortem(1)
        break
probe.sgpr_probe_queue_trace_end(buf, cap)
allw = np.frombuffer(buf, dtype=np.uint64, count=nwords)
wc = allw[8 * cap:8 * cap + 2048].reshape(-1, 2)
pc = allw[8 * cap + 2048:].reshape(512, 32, 4)
wc = wc[wc[:, 1] != 0]
t0 = int(wc[:, 1].min())
def place(x):
    x = x.astype(np.int64)
    return (x >> 32) & 0xF, (x >> 13) & 7, (x >> 8) & 0xF
xcc, se, cu = place(wc[:, 0])
cus = set(zip(xcc.tolist(), se.tolist(), cu.tolist()))
print("workers: %d workgroups on %d distinct CUs, per XCC %s, started within %.1f us" % (len(wc), len(cus), np.bincount(xcc, minlength=8).tolist(), (int(wc[:, 1].max()) - t0) / 100.0))
if failed:
    wst = (wc[:, 1].astype(np.int64) - t0) / 100.0
    print("  workers per (xcc, se):")
    for x in range(8):
        print("    xcc %d: %s   late workers (start > 20 us): %s" % (x, [int(((xcc == x) & (se == s_)).sum()) for s_ in range(8)],
              ["se%d cu%d @%.0f" % (se[i], cu[i], wst[i]) for i in range(len(wc)) if xcc[i] == x and wst[i] > 20]))
    x0 = pc[0][pc[0][:, 1] != 0]
    pxc, pse, pcu = place(x0[:, 0])
    print("  panel 0 workgroups at (xcc, se, cu), start:", ["(%d,%d,%d) @%.0f" % (pxc[i], pse[i], pcu[i], (int(x0[i, 1]) - t0) / 100.0) for i in range(len(x0))])
for k in range(8):
    m = pc[k][:, 1] != 0
    if not m.any():
        continue
    x = pc[k][m]
    st = (x[:, 1].astype(np.int64) - t0) / 100.0
    en = np.where(x[:, 2] != 0, (x[:, 2].astype(np.int64) - t0) / 100.0, -1)
    order = np.argsort(x[:, 3])
    xc, s_, c_ = place(x[:, 0])
    print("  panel %d: %d workgroups; strip:start-end(xcc): %s" % (k, m.sum(), " ".join("%d:%.0f-%.0f(%d)" % (int(x[o, 3]), st[o], en[o], xc[o]) for o in order)))

"""Gram-build time: the one-pair kernel (gram_pairs) vs the general-d kernel run at d = 1, back to
back and after an idle gap (what a bench step sees right after its barrier).
python tools/gram_speed.py [N]"""
import ctypes as C
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import _lib as L
from sympgpr_amd.dist import HipOps
from bench import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
q, P, z, hyp, s2 = synth(N)
n = 2 * N
ops = HipOps(torch.device("cuda", 0))
A = ops.empty(n * n)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
x, y = dev(q), dev(P)
X = dev(np.concatenate([q, P]))
def timeit(fn, gap=0.0):
    ts = []
    for _ in range(4):
        if gap:
            torch.cuda.synchronize()
            time.sleep(gap)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return ts
t1 = timeit(lambda: ops.gram_pairs("A", N, N, x, y, x, y, hyp, A, [0, N, N * n, N + N * n], n, L.G_ALL))
t2 = timeit(lambda: ops.gram_nd("A", 1, N, N, X, X, hyp, A, n))
for name, ts in (("gram_pairs", t1), ("gram_nd d=1", t2)):
    print("%-12s %s ms -> %.0f GB/s" % (name, ["%.2f" % t for t in ts], 8.0 * n * n / min(ts) / 1e6))
# how much of the time is the pair arithmetic?  the same build with the device-libs exp / sincos (~110 fp64 ops
# instead of ~50), and with families that need no sincos (C) or two exps (B)
for fam, flags, label in (("A", L.G_ALL | L.G_OCML, "A, OCML math"), ("C", L.G_ALL, "C (exp only)"), ("B", L.G_ALL, "B (two exps)"),
                          ("A", L.G_ALL | L.G_LOWER, "A, lower only")):
    ts = timeit(lambda: ops.gram_pairs(fam, N, N, x, y, x, y, hyp, A, [0, N, N * n, N + N * n], n, flags))
    print("%-16s %s ms" % (label, ["%.3f" % t for t in ts]))
for gap in (0.001, 0.05, 0.5):
    t1 = timeit(lambda: ops.gram_pairs("A", N, N, x, y, x, y, hyp, A, [0, N, N * n, N + N * n], n, L.G_ALL), gap)
    t2 = timeit(lambda: ops.gram_nd("A", 1, N, N, X, X, hyp, A, n), gap)
    print("idle %.3f s before each build: gram_pairs %s ms, gram_nd %s ms" % (gap, ["%.2f" % t for t in t1], ["%.2f" % t for t in t2]))

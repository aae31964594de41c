"""The k-per-launch cap experiment (SGPR_GEMM_KMAX, read once per process): one lower-triangular C -= A A^T through the
public entry sgpr_gemm_nt_dev, timed with events on the launch stream.
    SGPR_GEMM_KMAX=16384 python tools/probe_kmax.py 65536 65536 [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import _lib as L  # noqa: E402

m, k = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = L.load_library()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
A = torch.empty(m * k, dtype=torch.float64, device=dev)
for j0 in range(0, m * k, 1 << 28):           # fill in slabs (randn of 34 GB at once doubles the footprint)
    A[j0:j0 + (1 << 28)].normal_(generator=g)
Cm = torch.zeros(m * m, dtype=torch.float64, device=dev)
vp = lambda t: C.c_void_p(t.data_ptr())
ms = []
for r in range(reps + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.sgpr_gemm_nt_dev(m, m, k, -1.0, vp(A), m, vp(A), m, 1.0, vp(Cm), m, 1, 0, None))
    e1.record()
    torch.cuda.synchronize()
    if r:
        ms.append(e0.elapsed_time(e1))
flop = 2.0 * k * (m * (m + 1) / 2)
best = min(ms)
print("KMAX=%s m=%d k=%d lower: %s ms -> best %.2f TFLOP/s (algorithmic: on/below the diagonal)" % (
    os.environ.get("SGPR_GEMM_KMAX", "0"), m, k, " ".join("%.1f" % v for v in ms), flop / (best * 1e-3) / 1e12))

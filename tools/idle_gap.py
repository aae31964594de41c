"""How long may the device sit idle before the next heavy launch runs slow?  (the first launch of a block solve
behind milliseconds of idle device runs ~13 % slow for its whole 15 ms.)  One factorisation, then device-resident block solves
with a host sleep of G ms in front of each; prints the device time of the solve per gap.
    python tools/idle_gap.py [N=32768] [gaps in ms ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympgpr_amd.fit import SympFit
from bench import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
gaps = [float(v) for v in sys.argv[2:]] or [0.0, 0.1, 0.3, 1.0, 3.0, 10.0, 30.0, 100.0]
q, P, z, hyp, s2 = synth(N)
with SympFit("A", q, P, z, hyp, s2) as f:
    f.run()
    n = f.n
    dev = torch.device("cuda", torch.cuda.current_device())
    B0 = torch.randn((64, n), dtype=torch.float64, device=dev)
    Bd = torch.empty_like(B0)
    for _ in range(3):                                     # warm
        Bd.copy_(B0); torch.cuda.synchronize(); f.solve_rhs_dev(Bd.data_ptr(), 64)
    for g in gaps:
        ts = []
        for _ in range(4):
            Bd.copy_(B0)
            torch.cuda.synchronize()
            time.sleep(g * 1e-3)
            f.solve_rhs_dev(Bd.data_ptr(), 64)
            ts.append(f.solve_rhs_ms())
        print("n = %d, %6.1f ms of idle device in front of every solve: %s ms" % (n, g, " ".join("%.2f" % t for t in ts)), flush=True)

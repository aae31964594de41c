"""Many block solves in a row (sgpr_fit_solve_rhs_dev), every result compared bit for bit with the first one of its size: the
strip solves hand data from workgroup to workgroup inside a launch (data-tagged granules, flags, running partial sums); a
race or a lost hand-off shows as a differing bit, a timeout as an error of the call -- never as a hang.
    python tools/rhs_stress.py [--reps 300] [--nrhs 64] N [N ...]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympgpr_amd.fit import SympFit
from bench import synth

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--nrhs", type=int, default=64)
ap.add_argument("npts", type=int, nargs="+")
a = ap.parse_args()
for N in a.npts:
    q, P, z, hyp, s2 = synth(N)
    with SympFit("A", q, P, z, hyp, s2) as f:
        f.run()
        n = f.n
        dev = torch.device("cuda", torch.cuda.current_device())
        B0 = torch.randn((a.nrhs, n), dtype=torch.float64, device=dev)
        B0[0] = torch.from_numpy(z).to(dev)
        Bd = torch.empty_like(B0)
        ref = None
        t0 = time.perf_counter()
        bad = 0
        for r in range(a.reps):
            Bd.copy_(B0)
            torch.cuda.synchronize()
            f.solve_rhs_dev(Bd.data_ptr(), a.nrhs)
            if ref is None:
                ref = Bd.clone()
                al = f.alpha()
                err = float(np.linalg.norm(ref[0].cpu().numpy() - al) / np.linalg.norm(al))
            elif not torch.equal(Bd, ref):
                bad += 1
        print("n = %d, %d right-hand sides: %d solves in %.1f s, %d differ from the first; column 0 vs alpha %.1e"
              % (n, a.nrhs, a.reps, time.perf_counter() - t0, bad, err), flush=True)
        assert bad == 0 and err < 1e-10

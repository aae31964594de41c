"""The block-cyclic HIP driver with several ranks on ONE card (gloo carries the collectives): a scale check of
the multi-GPU path where only one GPU is available.  The parent never opens the GPU (a box allows 6 processes
on its card).  python tools/dist_one_card.py"""
import os
import socket
import sys
import time

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def problem(N):
    rng = np.random.default_rng(1234)
    q, P, z = rng.uniform(0, 2 * np.pi, N), rng.uniform(-3, 3, N), rng.standard_normal(2 * N)
    l = 2.0 * np.sqrt(12 * np.pi / N)
    return q, P, z, [l, l, 1.0], 1e-2 / l**2


def worker(rank, world, port, N, nb, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sympgpr_amd.dist import DistFit, HipOps
        q, P, z, hyp, s2 = problem(N)
        f = DistFit(HipOps(torch.device("cuda", 0)), "A", q, P, z, hyp, s2, nb=nb)
        torch.cuda.synchronize(); dist.barrier(); t0 = time.time()
        a = f.run().cpu().numpy().copy()
        torch.cuda.synchronize(); dist.barrier()
        out[rank] = (a, f.nll, time.time() - t0)
    finally:
        dist.destroy_process_group()


def single(rank, N, out):
    from sympgpr_amd.fit import SympFit
    q, P, z, hyp, s2 = problem(N)
    with SympFit("A", q, P, z, hyp, s2) as f:
        out["single"] = (f.run().alpha(), f.nll())


if __name__ == "__main__":
    for world, N, nb in [(4, 16384, 1024), (6, 12288, 1024)]:
        out = mp.Manager().dict()
        mp.spawn(worker, args=(world, free_port(), N, nb, out), nprocs=world, join=True)
        mp.spawn(single, args=(N, out), nprocs=1, join=True)
        a0, nll0, dt = out[0]
        a, nll = out["single"]
        same = all(np.array_equal(out[r][0], a0) for r in range(world))
        print("world %d N %d (n = %d) nb %d: %.1f s, ranks agree %s, |a - a_single|/|a| = %.2e, nll rel %.2e" % (
            world, N, 2 * N, nb, dt, same, np.linalg.norm(a0 - a) / np.linalg.norm(a), abs(nll0 - nll) / abs(nll)), flush=True)

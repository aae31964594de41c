"""bench.py's N > 1 leg: the 2-D block-cyclic fit across the GPUs of one node (one rank per
GPU, RCCL).  Strong scaling: the BASELINE problem (N = 65536 points, order n = 131072) is the
same at every GPU count."""
import json
import time

import numpy as np
import torch
import torch.distributed as dist

STAGE = ["start"]      # where this rank is (bench.py names it when a collective times out or a rank fails)


def run_distributed(args, rank, local_rank, world, synth, metric, peaks, synth_pairs=None, cpu_baseline=None):
    from .dist import DistFit, HipOps, grid_shape
    dev = torch.device("cuda", local_rank)
    ops = HipOps(dev)
    n_pts = args.n_pts
    d = getattr(args, "d", 1)
    n = 2 * d * n_pts
    pr, pc = grid_shape(world)
    nb = args.nb
    if d == 1:
        q, P, z, hyp, s2 = synth(n_pts)
        fit = DistFit(ops, args.family, q, P, z, hyp, s2, nb=nb)
        X = np.column_stack((q, P))
    else:
        X, z, hyp, s2 = synth_pairs(n_pts, d)
        fit = DistFit(ops, args.family, None, None, z, hyp, s2, nb=nb, X=X)
    nb = fit.nb      # the driver may have picked a smaller block size that divides N
    # every rank says what it is about to do before the first collective (stderr, one JSON line): its communicators, its HBM plan,
    # the collectives of the first panel steps in issue order -- if a run stops in RCCL, this is what the ranks disagree on
    import sys
    sys.stderr.write("sympgpr dist plan: " + json.dumps(fit.describe(2)) + "\n")
    sys.stderr.flush()

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def step(timed):
        if timed:
            ev[0].record()
        STAGE[0] = "gram build"
        fit.build()
        if timed:
            ev[1].record()
        STAGE[0] = "factor (panel exchange on the row / column communicators)"
        info = fit.factor()
        if info:
            raise np.linalg.LinAlgError("leading minor %d not positive definite" % info)
        if timed:
            ev[2].record()
        STAGE[0] = "solve (row / column reduces + world broadcasts)"
        fit.solve()
        if timed:
            ev[3].record()

    for _ in range(args.warmup):
        step(False)
    stage = np.zeros(3)
    STAGE[0] = "barrier before the timed steps"
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
        torch.cuda.synchronize()
        stage += np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
    barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    st = torch.tensor(stage / max(args.steps, 1), dtype=torch.float64, device=dev)
    dist.all_reduce(st, op=dist.ReduceOp.MAX)
    ms_per_step = float(dt.item()) / args.steps * 1e3
    my_stage = stage / max(args.steps, 1)
    stage = st.cpu().numpy()
    STAGE[0] = "gathering the per-rank records"
    # what every rank saw: device, PCI bus id, its own factor-stage time (the driver checks that N ranks really were
    # N devices, and how far apart they ran)
    props = torch.cuda.get_device_properties(dev)
    bus = getattr(props, "pci_bus_id", None)
    rec = [None] * world
    dist.all_gather_object(rec, {"rank": rank, "device_index": local_rank, "device": props.name,
                                 "pci_bus_id": int(bus) if bus is not None else None,
                                 "uuid": str(getattr(props, "uuid", "")), "factor_ms": float(my_stage[1])})
    cb = torch.tensor([float(fit.comm_bytes)], dtype=torch.float64, device=dev)   # bytes this rank received
    cb_max, cb_sum = cb.clone(), cb.clone()
    dist.all_reduce(cb_max, op=dist.ReduceOp.MAX)
    dist.all_reduce(cb_sum, op=dist.ReduceOp.SUM)

    # roofline pass: one more (untimed, collective) step with a HIP-event pair around every launch of the
    # MFMA kernel; rank 0's launches are reported
    prof = np.zeros(12)
    if not getattr(args, "no_launch_events", False):
        from . import _lib as L_
        L_.check(ops.lib.sgpr_profile_begin())
        step(False)
        torch.cuda.synchronize()
        L_.check(ops.lib.sgpr_profile_end(L_.dptr(prof)))
    # every rank's MFMA-kernel rate (all launches of its local trailing updates)
    # (launches of the 256 x 128 kernel, alone or overlapped, and of the 128 x 128 kernel small products go to: with the packed
    # block storage a trailing update is one product per local column block)
    n_mine, fl_mine, ms_mine = prof[0] + prof[3] + prof[8], prof[1] + prof[4] + prof[9], prof[2] + prof[5] + prof[10]
    mine = torch.tensor([fl_mine / (ms_mine * 1e-3) / 1e12 if ms_mine > 0 else 0.0], dtype=torch.float64, device=dev)
    ach_min, ach_max = mine.clone(), mine.clone()
    dist.all_reduce(ach_min, op=dist.ReduceOp.MIN)
    dist.all_reduce(ach_max, op=dist.ReduceOp.MAX)

    # a block of right-hand sides against the distributed factor (DistFit.solve_rhs; config 05_tokamak's multi-RHS predict when the
    # factor is spread over the grid) -- outside `value`, collective, and never fatal for the line: an error is reported as text
    nrhs = args.nrhs if getattr(args, "nrhs", -1) >= 0 else (64 if d == 3 else 16)
    rhs = None
    if nrhs > 0:
        STAGE[0] = "block of right-hand sides against the distributed factor"
        try:
            Bm = np.random.default_rng(5).standard_normal((n, nrhs))
            Bm[:, 0] = z
            Bt = torch.as_tensor(Bm).to(dev)
            barrier()
            t0 = time.perf_counter()
            Xs = fit.solve_rhs(Bt)
            barrier()
            t_rhs = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t_rhs, op=dist.ReduceOp.MAX)
            x0 = Xs[:, 0].contiguous()
            al = fit.alpha
            rhs = {"nrhs": nrhs, "ms": float(t_rhs.item()) * 1e3,
                   "column0_vs_alpha": float((torch.linalg.norm(x0 - al) / torch.linalg.norm(al)).item()),
                   "note": "wall time of DistFit.solve_rhs, max over ranks: 2 n / nb block steps, each one reduce + one broadcast of "
                           "nrhs x nb doubles and two local MFMA products"}
            del Xs, Bt
        except Exception as e:                 # (a collective failure here would hang the other ranks: the 5-minute timeout ends them)
            rhs = {"nrhs": nrhs, "error": "%s: %s" % (type(e).__name__, e)}

    # parity evidence: residual of Ky alpha = z on a sample of rows, rebuilt from the inputs
    a = fit.alpha.cpu().numpy()
    resid = None
    if rank == 0:
        m = min(n_pts, 512)
        idx = np.random.default_rng(0).choice(n_pts, m, replace=False)
        lib = ops.lib
        from . import _lib as L
        import ctypes as C
        dv = lambda v: torch.as_tensor(np.ascontiguousarray(v)).to(dev)
        D = 2 * d
        dXt = dv(np.asfortranarray(X[idx]).T.copy()).reshape(-1)     # (m x 2d) column-major, flat
        dXtr = dv(np.asfortranarray(X).T.copy()).reshape(-1)
        da = dv(a)
        out_t = torch.empty(m * D, dtype=torch.float64, device=dev)
        hyp_nd = L.f64(hyp)                                          # (lq.., lP.., sig): d = 1 is (lx, ly, sig)
        L.check(lib.sgpr_predict_nd_dev(L.family_id(args.family), d, m, C.c_void_p(dXt.data_ptr()), m, n_pts,
                                        C.c_void_p(dXtr.data_ptr()), n_pts, L.dptr(hyp_nd), len(hyp_nd),
                                        C.c_void_p(da.data_ptr()), C.c_void_p(out_t.data_ptr()), None))
        torch.cuda.synchronize()
        pred = out_t.cpu().numpy().reshape(D, m).T
        zz = z.reshape(D, n_pts).T[idx]
        aa = a.reshape(D, n_pts).T[idx]
        resid = float(np.linalg.norm(pred + s2 * aa - zz) / np.linalg.norm(zz))

    if rank == 0:
        chol_flop = n**3 / 3.0
        out = {
            "metric": metric,
            "value": chol_flop / (ms_per_step * 1e-3) / 1e12,
            "unit": "TFLOP/s (n^3/3 flop over the whole step: Gram build + Cholesky + solve)",
            "n_gpus": world, "rccl_world_size": dist.get_world_size(), "backend": dist.get_backend(),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" + (" -- REHEARSAL: all ranks on one card, collectives over gloo (not a multi-GPU measurement)"
                                   if dist.get_backend() == "gloo" else ""),
            "config": {"workload": "synthetic N=%d points, %s: matrix order n=%d (%.1f GB fp64) 2-D block-cyclic %dx%d, "
                                   "nb=%d, family %s" % (n_pts, "d=2 input coordinates (q,P)" if d == 1 else
                                                         "%d canonical pairs per point" % d, n, 8.0 * n * n / 1e9, pr,
                                                         pc, nb, args.family),
                       "n_pts": n_pts, "pairs_per_point": d, "order_n": n, "grid": [pr, pc], "nb": nb},
            "gram_gb_s": 8.0 * n * n / (stage[0] * 1e-3) / 1e9, "gram_ms": stage[0],
            "chol_tflops": chol_flop / (stage[1] * 1e-3) / 1e12, "chol_ms": stage[1],
            "solve_ms": stage[2], "residual_Ky_alpha_minus_z": resid, "nll": fit.nll,
            "ranks": rec,
            "factor_tflops_per_rank_min_max": [chol_flop / (max(r["factor_ms"] for r in rec) * 1e-3) / 1e12 / world,
                                               chol_flop / (min(r["factor_ms"] for r in rec) * 1e-3) / 1e12 / world],
            "panel_bytes_received_per_step": {"max_over_ranks": float(cb_max.item()), "sum_over_ranks": float(cb_sum.item()),
                                              "note": "row broadcast along the process row + column exchange down the "
                                                      "process column + L_KK and its leaf inverses; a rank receives "
                                                      "(1/pr + 1/pc) of every panel"},
        }
        n_l, fl, ms_l = n_mine, fl_mine, ms_mine
        if n_l > 0 and ms_l > 0:
            ach = fl / (ms_l * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_kernel<256,128> / <128,128> (rank 0's local trailing updates, one product per column block)",
                               "launches_256x128": int(prof[0] + prof[8]), "launches_128x128": int(prof[3]),
                               "achieved": ach, "peak": peaks["mfma"], "unit": "TFLOP/s", "frac": ach / peaks["mfma"],
                               "traffic": None, "traffic_source": None,
                               "timing": "one untimed extra step, HIP-event pair per launch on rank 0",
                               "launches": int(n_l), "flop_per_launch": fl / n_l, "avg_launch_ms": ms_l / n_l,
                               "achieved_min_max_over_ranks": [float(ach_min.item()), float(ach_max.item())],
                               "per_gpu_factor_stage_tflops": chol_flop / (stage[1] * 1e-3) / 1e12 / world}
        else:
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_kernel<256,128> (local trailing updates)",
                               "achieved": chol_flop / (stage[1] * 1e-3) / 1e12 / world, "peak": peaks["mfma"],
                               "unit": "TFLOP/s per GPU (factor stage incl. RCCL panel broadcasts)",
                               "frac": chol_flop / (stage[1] * 1e-3) / 1e12 / world / peaks["mfma"], "traffic": None}
        if rhs is not None:
            out["solve_rhs_distributed"] = rhs
        if cpu_baseline is not None and getattr(args, "cpu_sample", 0) > 0:
            # the same bounded CPU sample as the one-GPU line (rank 0 only; the other ranks wait at the barrier)
            cbl, _, _ = cpu_baseline(args.family if d == 1 else "A", args.cpu_sample)
            out["cpu_baseline"] = cbl
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()

#!/usr/bin/env python3
"""bench.py -- SympGPR training hot path on MI355X: Gram build + Cholesky + alpha.

One "step" = one pass of the hot path over one synthetic training set resident in HBM:
    Ky = build_K(x, x) + |sig2n| I   ->   L = cholesky(Ky)   ->   alpha = L^-T L^-1 z, nll
(the body of nll_chol, python/functions/func.py:189-196 of the reference).

Metric (BASELINE.json): "Gram-build GB/s + Cholesky fp64 TFLOP/s at N=65536 d=2;
|alpha-alpha_ref|/|alpha_ref|".  d = 2 input coordinates (q, P) per training point, so the
matrix order is n = 2N = 131072 (137 GB in place; see SURVEY.md 8 preamble).  `value` is the
whole-step Cholesky-equivalent rate (n^3/3 flop / wall time of the whole step, build and solve
included); the per-stage rates are reported beside it.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--n-pts N] [--d D] [--family A|B|C|D]
        N > 1: one rank per GPU over RCCL.  Under `python -m torch.distributed.run ...` (the driver's launcher)
        the ranks just join; typed by hand, `python bench.py --gpus N` starts them itself as child processes
        before anything in the parent touches the GPU.
        After the K timed steps ONE extra, untimed step records a HIP-event pair around every launch of the MFMA
        kernel (roofline.achieved); roofline.traffic comes from the committed PMC passes under profiles/ when
        their source hash matches the kernels being run.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Gram-build GB/s + Cholesky fp64 TFLOP/s at N=65536 d=2; |alpha-alpha_ref|/|alpha_ref|"
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
MFMA_F64_PEAK_TF = 78.6   # MI355X fp64 matrix peak, vendor spec


def synth_pairs(n_pts, d, seed=1234):
    """d canonical pairs per point: q_i ~ U(0, 2pi), P_i ~ U(-3, 3); l = 2 (12 pi)^(1/2) N^(-1/(2d)) keeps
    about the same number of neighbours within a length scale as the d = 1 setting."""
    rng = np.random.default_rng(seed)
    X = np.column_stack([rng.uniform(0, 2 * np.pi, (n_pts, d)), rng.uniform(-3, 3, (n_pts, d))])
    z = rng.standard_normal(2 * d * n_pts)
    l = 2.0 * np.sqrt(12 * np.pi) * n_pts ** (-1.0 / (2 * d))
    return X, z, np.append(np.full(2 * d, l), 1.0), 1e-2 / l**2


def synth(n_pts, seed=1234):
    """q ~ U(0, 2pi), P ~ U(-3, 3), z ~ N(0,1); l = 2 sqrt(12 pi / N), sig = 1,
    sig2n = 1e-2 / l^2 (bounded condition number)."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(0, 2 * np.pi, n_pts)
    P = rng.uniform(-3, 3, n_pts)
    z = rng.standard_normal(2 * n_pts)
    l = 2.0 * np.sqrt(12 * np.pi / n_pts)
    return q, P, z, np.array([l, l, 1.0]), 1e-2 / l**2


def cpu_baseline(family, n_pts_sample):
    """The reference's own Fortran build_K (oracle/_ref, 1 thread -- the reference is
    single-threaded) + the SciPy calls of func.py:193-194, on a bounded sample of the same
    synthetic workload.  Falls back to the C port (oracle/liboracle.so) when _ref is absent."""
    import scipy.linalg
    from oracle.oracle import Oracle, Ref
    q, P, z, hyp, s2 = synth(n_pts_sample)
    n = 2 * n_pts_sample
    kind = "reference" if (Ref.available() and family in "AC") else "port"
    t0 = time.perf_counter()
    if kind == "reference":
        K = Ref().build_K(family, q, P, q, P, hyp)
    else:
        K = Oracle().build_K(family, q, P, q, P, hyp, threads=1)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    K[np.diag_indices(n)] += abs(s2)
    Lf = scipy.linalg.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    t_chol = time.perf_counter() - t0
    t0 = time.perf_counter()
    alpha = scipy.linalg.solve_triangular(
        Lf.T, scipy.linalg.solve_triangular(Lf, z, lower=True, check_finite=False),
        lower=False, check_finite=False)
    t_solve = time.perf_counter() - t0
    total = t_build + t_chol + t_solve
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    # BASELINE.md 4's other two figures, on smaller samples so that the leg stays bounded: build_K on all cores (the C port of the
    # same loops, oracle/sympgpr_oracle.c, OpenMP over columns -- the reference itself is single-threaded) and the factor /
    # solve with ONE BLAS thread
    variants = {}
    try:
        ncores = os.cpu_count() or 1
        t0 = time.perf_counter()
        Oracle().build_K(family, q, P, q, P, hyp, threads=ncores)
        tb = time.perf_counter() - t0
        variants["build_all_cores"] = {"kind": "port", "threads": ncores, "n": n, "seconds": tb, "gb_s": 8.0 * n * n / tb / 1e9}
    except Exception as e:                       # the port is test infrastructure: its absence must not break the bench line
        variants["build_all_cores"] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:
        from threadpoolctl import threadpool_limits
        n1 = min(n, 6144)
        K1 = np.array(K[:n1, :n1], order="F")    # (K holds L now; any SPD matrix of that order times the same)
        K1 = np.tril(K1) @ np.tril(K1).T + n1 * np.eye(n1)
        with threadpool_limits(limits=1):
            t0 = time.perf_counter()
            L1 = scipy.linalg.cholesky(K1, lower=True, overwrite_a=True, check_finite=False)
            tc1 = time.perf_counter() - t0
            t0 = time.perf_counter()
            scipy.linalg.solve_triangular(L1.T, scipy.linalg.solve_triangular(L1, z[:n1], lower=True, check_finite=False),
                                          lower=False, check_finite=False)
            ts1 = time.perf_counter() - t0
        variants["factor_solve_1_blas_thread"] = {"n": n1, "cholesky_seconds": tc1, "cholesky_gflops": n1**3 / 3.0 / tc1 / 1e9,
                                                  "two_solves_seconds": ts1}
    except Exception as e:
        variants["factor_solve_1_blas_thread"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # the host (nproc, CPU model, BLAS), the two smaller sizes of the plan, and the full-size figures
    # EXTRAPOLATED from the n = 16384 sample (build ~ n^2, factor ~ n^3) -- labelled as such
    host = {"nproc": os.cpu_count()}
    try:
        with open("/proc/cpuinfo") as fh:
            host["cpu_model"] = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), None)
    except OSError:
        host["cpu_model"] = None
    try:
        from threadpoolctl import threadpool_info
        host["blas"] = [{k: p.get(k) for k in ("internal_api", "version", "num_threads", "threading_layer")} for p in threadpool_info()
                        if p.get("user_api") == "blas"]
    except Exception:
        host["blas"] = None
    sizes = {}
    for ns in (512, 2048):
        try:
            qs_, Ps_, zs_, hs_, s2s_ = synth(ns)
            m_ = 2 * ns
            t0 = time.perf_counter()
            Ks = Ref().build_K(family, qs_, Ps_, qs_, Ps_, hs_) if kind == "reference" else Oracle().build_K(family, qs_, Ps_, qs_, Ps_, hs_, threads=1)
            tb_ = time.perf_counter() - t0
            Ks[np.diag_indices(m_)] += abs(s2s_)
            t0 = time.perf_counter()
            scipy.linalg.cholesky(Ks, lower=True, overwrite_a=True, check_finite=False)
            tc_ = time.perf_counter() - t0
            sizes["n=%d" % m_] = {"build_seconds_1_thread": tb_, "build_gb_s": 8.0 * m_ * m_ / tb_ / 1e9, "cholesky_seconds": tc_,
                                  "cholesky_gflops": m_**3 / 3.0 / tc_ / 1e9}
        except Exception as e:
            sizes["n=%d" % (2 * ns)] = {"error": "%s: %s" % (type(e).__name__, e)}
    return {
        "host": host, "smaller_sizes": sizes,
        "value": (n**3 / 3.0) / total / 1e12, "unit": "TFLOP/s", "cores": int(blas_threads),
        "kind": kind,
        "sample": "n=%d (N=%d pts): build_K 1 thread %.2fs = %.3f GB/s; scipy cholesky %d threads %.2fs = "
                  "%.1f GFLOP/s; 2x solve_triangular %.2fs" % (n, n_pts_sample, t_build, 8.0 * n * n / t_build / 1e9,
                                                               blas_threads, t_chol, n**3 / 3.0 / t_chol / 1e9, t_solve),
        "gram_gb_s": 8.0 * n * n / t_build / 1e9, "chol_tflops": n**3 / 3.0 / t_chol / 1e12,
        "host_cpus": os.cpu_count(), "variants": variants,
    }, alpha, (q, P, z, hyp, s2)


def kernel_code_hash():
    """Identifies the kernel sources a PMC traffic profile belongs to (bench.py reports `traffic` only
    from a profile taken with exactly these sources)."""
    import hashlib
    h = hashlib.sha1()
    for f in ("common.h", "devmath.h", "pair_eval.h", os.path.join("generated", "pair_generated.h"), "leaf.h", "gemm_f64.hip",
              "gemm_tile.h", "chol.hip", "trsv.hip", "gram.hip", "gram_nd.hip", "blas_small.hip", "capi.hip"):
        # (cholq.hip / cholq.h are not in: the task-queue driver runs orders 13312 .. 28672 only, the traffic profiles are of
        # n >= 98304, where none of its code is reached)
        with open(os.path.join(ROOT, "sympgpr_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def load_traffic(n_pts, d, family, lower_only):
    """HBM-side bytes per launch from the committed separate `rocprofv3 --pmc` passes of this same
    configuration (newest profiles/rNN first).  Returns ({}, None) when there is none or when the
    kernels have changed since it was taken."""
    import glob
    code = kernel_code_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic_*.json")), reverse=True):
        try:
            pm = json.load(open(path))
        except Exception:
            continue
        c = pm.get("config", {})
        if (c.get("n_pts") == n_pts and c.get("family") == family and c.get("pairs_per_point", 1) == d
                and c.get("triangle", "full") == ("lower" if lower_only else "full") and pm.get("code_hash") == code):
            out = {}
            for key, name in (("gemm", "gemm_nt_kernel<256, 128>"), ("gram", "gram_pairs_kernel"), ("gram", "gram_nd_kernel")):
                if name in pm:
                    out[key] = pm[name]["traffic_bytes_per_launch"]
            return out, "%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, kernels %s)" % (
                os.path.relpath(path, ROOT), code)
    return {}, None


def batch_leg(args):
    """python bench.py --batch ORDER,COUNT: COUNT independent fits of order ORDER (the body of nll_chol, python/functions/func.py:
    189-196, per problem: Gram build, Cholesky, two solves, nll) through ONE sgpr_fit_batch call -- the reference's only batch axis
    (05_tokamak/Split_SympGPR/main.py:36-41,63-66).  Wall time of the call (inputs from host arrays through the pinned staging
    block, outputs back), median of the timed calls."""
    from sympgpr_amd.fit import fit_batch
    n, B = (int(v) for v in args.batch.split(","))
    Np = n // 2
    rng = np.random.default_rng(3)
    x, y = rng.uniform(0, 2 * np.pi, (B, Np)), rng.uniform(-3, 3, (B, Np))
    z = rng.standard_normal((B, n))
    l = 2.0 * np.sqrt(12 * np.pi / Np)
    hyp = np.tile([l, l, 1.0], (B, 1))
    s2 = np.full(B, 1e-2 / l**2)
    for _ in range(max(args.warmup, 1)):
        fit_batch(args.family, x, y, z, hyp, s2, want_alpha=False)
    ts = []
    for _ in range(max(args.steps, 5)):
        t0 = time.perf_counter()
        _, nll, info = fit_batch(args.family, x, y, z, hyp, s2, want_alpha=False)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    flop_fit = n**3 / 3.0 + 2.0 * n * n            # factor + the two triangular solves (the Gram build is not flop on the matrix cores)
    out = {"metric": "batched nll_chol bodies per second (one sgpr_fit_batch call per batch)", "value": B / t, "unit": "fits/s",
           "n_gpus": 1, "steps": len(ts), "warmup": max(args.warmup, 1), "ms_per_step": t * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "%d independent fits of matrix order %d (N = %d points each), family %s" % (B, n, Np, args.family),
                      "order_n": n, "batch": B},
           "us_per_fit": t / B * 1e6, "tflops": B * flop_fit / t / 1e12,
           "roofline": {"bound": "mfma", "kernel": "fit_batch_kernel (order <= 256) / mid_build + panel_batch + mid_syrk + mid_solve kernels",
                        "achieved": B * flop_fit / t / 1e12, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s",
                        "frac": B * flop_fit / t / 1e12 / MFMA_F64_PEAK_TF, "traffic": None,
                        "flop_per_fit": flop_fit, "timing": "host wall time of the whole call, median of %d" % len(ts)},
           "all_info_zero": bool(np.all(np.asarray(info) == 0))}
    if args.cpu_sample > 0:
        import scipy.linalg
        from oracle.oracle import Oracle, Ref
        kind = "reference" if (Ref.available() and args.family in "AC") else "port"
        bk = Ref().build_K if kind == "reference" else (lambda *a: Oracle().build_K(*a, threads=1))
        m, t0, vals = 0, time.perf_counter(), []
        while m < B and time.perf_counter() - t0 < 10.0:
            K = bk(args.family, x[m], y[m], x[m], y[m], hyp[m])
            K[np.diag_indices(n)] += s2[m]
            Lf = scipy.linalg.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
            al = scipy.linalg.solve_triangular(Lf.T, scipy.linalg.solve_triangular(Lf, z[m], lower=True, check_finite=False),
                                               lower=False, check_finite=False)
            vals.append(0.5 * z[m] @ al + np.sum(np.log(Lf.diagonal())))
            m += 1
        tc = (time.perf_counter() - t0) / m
        out["cpu_baseline"] = {"value": 1.0 / tc, "unit": "fits/s", "cores": 1, "kind": kind,
                               "sample": "%d of the %d problems, one after the other: the reference's build_K + scipy cholesky / "
                                         "solve_triangular (BLAS threads as the box gives them)" % (m, B)}
        out["nll_rel_err_vs_cpu"] = float(np.max(np.abs(np.asarray(nll[:m]) - np.asarray(vals)) / np.abs(vals)))
    print(json.dumps(out))


def launch_ranks(n):
    """`python bench.py --gpus N` typed by hand: start the N ranks as fresh child processes BEFORE
    anything in this process touches the GPU (a process that has initialised HIP must never be
    replaced or forked into ranks).  Returns the children's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    # the launcher's own argparse resolves abbreviations before it hands the rest to the script: "--d" is an ambiguous
    # prefix of its --duplicate-* options, so the children get the long spelling
    for a in sys.argv[1:]:
        cmd.append("--pairs-per-point" + a[3:] if (a == "--d" or a.startswith("--d=")) else a)
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-pts", type=int, default=65536, help="training points N (matrix order n = 2*d*N)")
    ap.add_argument("--family", default="A")
    ap.add_argument("--d", "--pairs-per-point", dest="d", type=int, default=1,
                    help="canonical pairs per training point (matrix order n = 2*d*N); 1 = the reference's "
                         "(q, P) layout, 2 / 3 = BASELINE configs 03_henon_heiles / 05_tokamak; N = 65536 with "
                         "--d 2 is the n = 262144 multi-GPU configuration")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="N of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--nb", type=int, default=2048, help="block size of the multi-GPU block-cyclic layout")
    ap.add_argument("--force-dist", action="store_true", help="run the block-cyclic driver even on 1 GPU")
    ap.add_argument("--no-launch-events", action="store_true",
                    help="skip the extra untimed pass that times every GEMM launch with HIP events")
    ap.add_argument("--nrhs", type=int, default=-1,
                    help="right-hand sides of the extra, separately timed block solve X = L^-T L^-1 B with the cached factor "
                         "(BASELINE config 05_tokamak: multi-RHS predict TRSM); default 64 with --d 3, else 0 = skip")
    ap.add_argument("--batch", default="",
                    help="ORDER,COUNT: instead of the big fit, time COUNT independent nll_chol bodies of matrix order ORDER in one "
                         "batched call (sgpr_fit_batch: a CMA-ES generation / the Split_SympGPR sections) and print one JSON line "
                         "with fits/s, TFLOP/s and the roofline fraction, the reference's CPU path beside it")
    ap.add_argument("--cond-iters", type=int, default=30,
                    help="power / inverse iteration steps of the condition estimate printed beside the parity numbers (0 = skip)")
    ap.add_argument("--lower-only", action="store_true",
                    help="build only the lower triangle of K (what the factor reads) instead of the "
                         "full matrix build_K defines")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))     # nothing above has touched the GPU

    import torch
    import sympgpr_amd
    from sympgpr_amd import _lib as L
    from sympgpr_amd.fit import SympFit

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hook for a one-GPU box (not a measurement): SGPR_BENCH_ONE_CARD=1 puts every rank on cuda:0 and
    # carries the collectives over gloo, so that the whole N > 1 leg (launcher, barriers, timing reduction,
    # per-launch pass, JSON line) can be exercised without a second GPU; at most 6 ranks share a card
    one_card = os.environ.get("SGPR_BENCH_ONE_CARD") == "1"
    if one_card:
        local_rank = 0
    if not one_card and sympgpr_amd.device_count() < max(1, min(args.gpus, world)):
        raise SystemExit("bench.py --gpus %d needs %d GPU(s), this box shows %d: libsympgpr_hip.so has no CPU fallback"
                         % (args.gpus, args.gpus, sympgpr_amd.device_count()))
    if local_rank >= sympgpr_amd.device_count():
        raise SystemExit("rank %d: needs %d GPUs, this box shows %d" % (rank, world, sympgpr_amd.device_count()))
    torch.cuda.set_device(local_rank)
    lib = L.load_library()
    L.check(lib.sgpr_set_device(local_rank))
    if world > 1:
        import torch.distributed as dist
        from datetime import timedelta
        # a collective that does not complete within 5 minutes ends the rank with a non-zero exit code (the NCCL / RCCL
        # watchdog aborts the process; nothing here re-launches or re-execs a rank that has touched the GPU)
        if one_card:
            dist.init_process_group("gloo", timeout=timedelta(minutes=5))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=timedelta(minutes=5))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1 or args.force_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            import torch.distributed as dist
            from datetime import timedelta
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank),
                                    timeout=timedelta(minutes=5))
        from sympgpr_amd.dist_bench import run_distributed
        try:
            return run_distributed(args, rank, local_rank, world, synth, METRIC,
                                   {"mfma": MFMA_F64_PEAK_TF, "hbm": HBM_PEAK_GBS}, synth_pairs, cpu_baseline)
        except Exception as e:           # a collective timeout, a factor error on some rank, ...: say where, exit non-zero
            from sympgpr_amd import dist_bench
            sys.stderr.write("bench.py rank %d/%d FAILED in stage '%s': %s: %s\n" % (rank, world, dist_bench.STAGE[0], type(e).__name__, e))
            sys.stderr.flush()
            os._exit(3)

    if args.batch:
        return batch_leg(args)

    n_pts = args.n_pts
    d = args.d
    n = 2 * d * n_pts
    if d == 1:
        q, P, z, hyp, s2 = synth(n_pts)
        X = np.column_stack((q, P))
        fit = SympFit(args.family, q, P, z, hyp, s2, lower_only=args.lower_only)
    else:
        X, z, hyp, s2 = synth_pairs(n_pts, d)
        fit = SympFit.pairs(args.family, X, z, hyp, s2)

    for _ in range(args.warmup):
        fit.run()
    stage = np.zeros(3)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fit.run()
        stage += np.array(fit.stage_ms())
    barrier()
    dt = time.perf_counter() - t0
    stage /= max(args.steps, 1)
    ms_per_step = dt / args.steps * 1e3

    # roofline pass: ONE more step, outside the timed region, with a HIP-event pair (from a fixed
    # pool) around every launch of the MFMA kernel on the stream it is launched on
    prof = np.zeros(12)
    if not args.no_launch_events:
        L.check(lib.sgpr_profile_begin())
        fit.run()
        torch.cuda.synchronize()
        L.check(lib.sgpr_profile_end(L.dptr(prof)))

    # parity evidence at full size: Ky alpha == z through the device's K*-row kernel (the host-side
    # re-evaluation of those rows by the oracle is part of the cpu_baseline leg below)
    a = fit.alpha()
    m = min(n_pts, 2048)
    idx = np.random.default_rng(0).choice(n_pts, m, replace=False)
    if d == 1:
        op, oq = fit.predict_rows(q[idx], P[idx])
        r = np.concatenate([op + s2 * a[idx] - z[idx], oq + s2 * a[n_pts + idx] - z[n_pts + idx]])
        resid = float(np.linalg.norm(r) / np.linalg.norm(np.concatenate([z[idx], z[n_pts + idx]])))
    else:
        pred = fit.predict_pairs(X[idx])                       # (m, 2d): K alpha at training points
        zz = z.reshape(2 * d, n_pts).T[idx]
        aa = a.reshape(2 * d, n_pts).T[idx]
        resid = float(np.linalg.norm(pred + s2 * aa - zz) / np.linalg.norm(zz))
    nll = fit.nll()
    # BASELINE config 05_tokamak's other half: a block of right-hand sides against the cached factor (what the reference does with
    # matmul(Kyinv, ztrain) per prediction, sympgpr.f90:72,85,121).  Timed by HIP events around the device part (B already in HBM),
    # outside the step `value` is computed from.
    nrhs = args.nrhs if args.nrhs >= 0 else (64 if d == 3 else 0)
    rhs = None
    if nrhs > 0:
        import torch
        Bm = np.random.default_rng(5).standard_normal((n, nrhs))
        Bm[:, 0] = z
        # (a) right-hand sides resident in HBM (sgpr_fit_solve_rhs_dev), four solves back to back: this is the figure.  The first
        # of them follows host work and is reported but not averaged: the first heavy launch behind an idle device runs ~13 % slow
        # (lower shader clock AND more cycles per tile for the whole 15 ms).
        dev = torch.device("cuda", torch.cuda.current_device())
        B0 = torch.from_numpy(np.ascontiguousarray(Bm.T)).to(dev)      # (nrhs, n) contiguous = n x nrhs column-major
        Bd = torch.empty_like(B0)
        ts = []
        for _ in range(4):
            Bd.copy_(B0)
            torch.cuda.synchronize()
            fit.solve_rhs_dev(Bd.data_ptr(), nrhs)
            ts.append(fit.solve_rhs_ms())
        x0 = Bd[0].cpu().numpy()
        # (b) the host-buffer entry (sgpr_fit_solve_rhs: B copied in and out around the same device work), for the record
        th = []
        for _ in range(2):
            Xs = fit.solve_rhs(Bm)
            th.append(fit.solve_rhs_ms())
        del B0, Bd
        rhs = {"nrhs": nrhs, "ms": float(np.mean(ts[1:])), "ms_all": [float(v) for v in ts],
               "ms_device_part_of_host_buffer_calls": [float(v) for v in th],
               "column0_vs_alpha": float(np.linalg.norm(x0 - a) / np.linalg.norm(a)),
               "column0_vs_alpha_host_buffers": float(np.linalg.norm(Xs[:, 0] - a) / np.linalg.norm(a))}
    # cond_2(Ky) of the full-size matrix, from below, with the device's own kernels (beside every parity number)
    cond = fit.cond_estimate(args.cond_iters) if args.cond_iters > 0 else None
    # the Gram kernel alone, back to back (outside the timed region, not part of `value`): the build
    # inside a step starts on an idle chip right after the barrier and carries that warm-up
    rep = []
    for _ in range(4):
        fit.build()
        rep.append(fit.stage_ms()[0])
    fit.close()

    gram_bytes = 8.0 * n * (n + 1) / 2 if args.lower_only else 8.0 * n * n
    chol_flop = n**3 / 3.0
    out = {
        "metric": METRIC,
        "value": chol_flop / (ms_per_step * 1e-3) / 1e12,
        "unit": "TFLOP/s (n^3/3 flop over the whole step: Gram build + Cholesky + solve)",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "synthetic N=%d points, %s: matrix order n=%d (%.1f GB fp64), family %s, "
                               "sig2n=1e-2/l^2" % (n_pts, "d=2 input coordinates (q,P), l=2*sqrt(12pi/N)" if d == 1 else
                                                   "%d canonical pairs per point" % d, n, 8.0 * n * n / 1e9, args.family),
                   "n_pts": n_pts, "pairs_per_point": d, "order_n": n,
                   "triangle": "lower" if args.lower_only else "full"},
        "gram_gb_s": gram_bytes / (stage[0] * 1e-3) / 1e9,
        "gram_ms": stage[0],
        "gram_repeat_ms": min(rep[1:]),
        "gram_repeat_gb_s": (8.0 * n * (n + 1) / 2 if args.lower_only else 8.0 * n * n) / (min(rep[1:]) * 1e-3) / 1e9,
        "chol_tflops": chol_flop / (stage[1] * 1e-3) / 1e12,
        "chol_ms": stage[1],
        "solve_ms": stage[2],
        "solve_gb_s": 8.0 * n * n / (stage[2] * 1e-3) / 1e9,
        "residual_Ky_alpha_minus_z": resid,
        "nll": nll,
        "cond_estimate": cond,
    }
    traffic, traffic_source = load_traffic(n_pts, d, args.family, args.lower_only)
    alone_n, alone_flop, alone_ms = prof[0], prof[1], prof[2]
    ov_n, ov_flop, ov_ms = prof[8], prof[9], prof[10]
    if alone_n + ov_n > 0:
        # `achieved` = ALL launches of the kernel: sum of their algorithmic flop / sum of their durations -- what the
        # rocprofv3 kernel stats of the same command reproduce.  The launches that had the device to themselves
        # (the look-ahead driver's run two streams at once, so theirs overlap other work) are quoted beside it.
        all_n, all_flop, all_ms = alone_n + ov_n, alone_flop + ov_flop, alone_ms + ov_ms
        ach = all_flop / (all_ms * 1e-3) / 1e12
        ach_alone = alone_flop / (alone_ms * 1e-3) / 1e12 if alone_ms > 0 else None
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_kernel<256,128> (fp64 MFMA trailing update)",
                           "achieved": ach, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s",
                           "frac": ach / MFMA_F64_PEAK_TF, "traffic": traffic.get("gemm"),
                           "traffic_source": traffic_source,
                           "timing": "one untimed extra step, HIP-event pair per launch on the launch stream; all launches",
                           "launches": int(all_n), "flop_per_launch": all_flop / all_n,
                           "avg_launch_ms": all_ms / all_n,
                           "achieved_alone": ach_alone,
                           "frac_alone": ach_alone / MFMA_F64_PEAK_TF if ach_alone else None,
                           "launches_alone": int(alone_n), "launches_overlapped": int(ov_n),
                           "achieved_overlapped": (ov_flop / (ov_ms * 1e-3) / 1e12) if ov_ms > 0 else None,
                           "sum_launch_ms_alone": alone_ms, "sum_launch_ms_overlapped": ov_ms,
                           "launches_untimed": int(prof[11]),
                           "largest_launch_tflops": prof[6] / (prof[7] * 1e-3) / 1e12 if prof[7] > 0 else None}
    elif 13312 <= n <= 28672 and n % 256 == 0 and os.environ.get("SGPR_POTRF_Q", "1") != "0":
        # no launch of the grid-wide MFMA kernel in this factorisation: orders 13312 .. 28672 run every trailing update inside the
        # persistent worker kernel of the task-queue Cholesky (several instances are enqueued, all but the first
        # normally find nothing to do).  Its duration is the factor stage; the flop are the whole factorisation's (the panel
        # kernel's share, the 128 x 128 leaves and their rows, is < 2 % at these orders and is counted in).
        ach = out["chol_tflops"]
        out["roofline"] = {"bound": "mfma", "kernel": "chol_queue_kernel + panel_seq_kernel (task-queue Cholesky: every trailing update and "
                                                      "rows-below solve in one persistent worker grid, same 256x128 fp64 MFMA body)",
                           "achieved": ach, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F64_PEAK_TF,
                           "traffic": None, "traffic_source": None,
                           "timing": "HIP events around the factor stage of the timed steps (the persistent kernels span the stage)",
                           "launches": None, "flop_per_launch": None, "flop": n**3 / 3.0, "stage_ms": out["chol_ms"]}
    else:
        # small orders / SGPR_POTRF_Q=0 runs without a single timed launch of the 256x128 kernel: the stage figure, labelled as such
        ach = out["chol_tflops"]
        out["roofline"] = {"bound": "mfma", "kernel": "factor stage as a whole (no launch of gemm_nt_kernel<256,128> was timed at this order)",
                           "achieved": ach, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F64_PEAK_TF,
                           "traffic": None, "traffic_source": None, "timing": "HIP events around the factor stage of the timed steps",
                           "launches": None, "flop_per_launch": None, "flop": n**3 / 3.0, "stage_ms": out["chol_ms"]}
    out["roofline_gram"] = {"bound": "hbm", "kernel": "gram_pairs_kernel" if d == 1 else "gram_nd_kernel", "achieved": out["gram_gb_s"],
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": out["gram_gb_s"] / HBM_PEAK_GBS,
                            "traffic": traffic.get("gram"), "traffic_source": traffic_source,
                            "achieved_back_to_back": out["gram_repeat_gb_s"],
                            "frac_back_to_back": out["gram_repeat_gb_s"] / HBM_PEAK_GBS}
    if rhs:
        passes = (rhs["nrhs"] + 63) // 64                 # L is streamed once per triangular solve per 64 columns
        l_bytes = 8.0 * n * n * passes                    # 4 n^2 B per solve, two solves
        flop = 2.0 * n * n * rhs["nrhs"]
        gbs, tf = l_bytes / (rhs["ms"] * 1e-3) / 1e9, flop / (rhs["ms"] * 1e-3) / 1e12
        floor_hbm, floor_mfma = l_bytes / HBM_PEAK_GBS / 1e6, flop / MFMA_F64_PEAK_TF / 1e9      # ms
        out["solve_rhs"] = dict(rhs, l_read_gb_s=gbs, tflops=tf)
        out["roofline_solve_rhs"] = {
            "kernel": "trsm_strips_kernel (one launch per triangular solve: L streamed once per 64 right-hand sides)",
            "bound": "mfma" if floor_mfma > floor_hbm else "hbm",
            "achieved": tf if floor_mfma > floor_hbm else gbs, "peak": MFMA_F64_PEAK_TF if floor_mfma > floor_hbm else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if floor_mfma > floor_hbm else "GB/s",
            "frac": (tf / MFMA_F64_PEAK_TF) if floor_mfma > floor_hbm else (gbs / HBM_PEAK_GBS),
            "frac_hbm": gbs / HBM_PEAK_GBS, "frac_mfma": tf / MFMA_F64_PEAK_TF,
            "floor_ms_hbm": floor_hbm, "floor_ms_mfma": floor_mfma, "traffic": None,
            "algorithmic": "8 n^2 B read of L per forward + backward pair per 64 columns; 2 n^2 nrhs flop",
            "timing": "HIP events around the pack, two solve launches and unpack of sgpr_fit_solve_rhs_dev (right-hand sides resident in HBM); mean of the last 3 of 4 back-to-back calls"}
    if args.cpu_sample > 0:
        cb, a_ref, (qs, Ps, zs, hs, s2s) = cpu_baseline(args.family if d == 1 else "A", args.cpu_sample)
        with SympFit(args.family if d == 1 else "A", qs, Ps, zs, hs, s2s) as fs:
            a_gpu = fs.run().alpha()
            cond_s = fs.cond_estimate(args.cond_iters) if args.cond_iters > 0 else None
        err1 = float(np.linalg.norm(a_gpu - a_ref) / np.linalg.norm(a_ref))
        if d == 1:
            out["alpha_rel_err"] = err1
            out["alpha_rel_err_at"] = "n=%d, family %s, vs the CPU baseline's solve" % (2 * args.cpu_sample, args.family)
            out["alpha_rel_err_cond_estimate"] = cond_s
        else:
            # the configuration's OWN kernel: a sample of the same family and d against the oracle's fit (oracle.fit_nd: restated
            # kernels + SciPy's cholesky / solve_triangular); the d = 1 family-A figure of the CPU baseline's sample beside it
            from oracle.oracle import Oracle
            ns = max(64, min(n_pts, 12288 // (2 * d)))
            Xs, zs2, hs2, s2s2 = synth_pairs(ns, d)
            a_o, _, _ = Oracle().fit_nd(args.family, Xs, zs2, hs2, s2s2)
            with SympFit.pairs(args.family, Xs, zs2, hs2, s2s2) as fs:
                a_g = fs.run().alpha()
                cond_s2 = fs.cond_estimate(args.cond_iters) if args.cond_iters > 0 else None
            out["alpha_rel_err"] = float(np.linalg.norm(a_g - a_o) / np.linalg.norm(a_o))
            out["alpha_rel_err_at"] = "n=%d (N=%d points, %d canonical pairs each), family %s, vs oracle.fit_nd" % (2 * d * ns, ns, d, args.family)
            out["alpha_rel_err_cond_estimate"] = cond_s2
            out["alpha_rel_err_d1_family_A"] = err1
            out["alpha_rel_err_d1_family_A_at"] = "n=%d vs the CPU baseline's solve" % (2 * args.cpu_sample)
        n_s = 2 * args.cpu_sample
        cb["extrapolated_to_full_size"] = {
            "order_n": n, "from": "the n = %d sample above: build ~ n^2 (1 thread), cholesky ~ n^3 (%d BLAS threads)" % (n_s, cb["cores"]),
            "build_seconds": (8.0 * n_s * n_s / cb["gram_gb_s"] / 1e9) * (n / n_s) ** 2,
            "cholesky_seconds": ((n_s**3 / 3.0) / cb["chol_tflops"] / 1e12) * (n / n_s) ** 3,
            "label": "EXTRAPOLATED, not measured"}
        out["cpu_baseline"] = cb
        # the checker's second job: rows of Ky at full size re-evaluated on the host by the oracle
        # (restated Fortran formulas), so the full-size residual does not rest on any device formula
        from oracle.oracle import Oracle
        orc = Oracle()
        ids = idx[:128]
        if d == 1:
            Krows = orc.build_K(args.family, q[ids], P[ids], q, P, hyp, threads=min(8, os.cpu_count() or 1))
        else:
            Krows = orc.build_K_nd(args.family, X[ids], X, hyp)
        rows = np.concatenate([b * n_pts + ids for b in range(2 * d)])
        rr = Krows @ a + s2 * a[rows] - z[rows]
        out["residual_oracle_rows"] = float(np.linalg.norm(rr) / np.linalg.norm(z[rows]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()

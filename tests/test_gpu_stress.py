"""Randomised sizes through the host-level factor / solve / Gram entry points: ragged tiles, every
driver switch-over (leaf, look-ahead block sizes, recursion), the triangular tile map."""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


def test_random_sizes_factor_and_solve():
    from sympgpr_amd import ops
    rng = np.random.default_rng(2026)
    sizes = sorted(set([1, 2, 127, 128, 129, 255, 256, 257, 511, 512, 513, 640, 1023, 1025, 2047, 2049, 4097] +
                       list(rng.integers(3, 6000, 28))))
    for n in sizes:
        B = rng.standard_normal((n, n + 3))
        A = B @ B.T / n + 0.5 * np.eye(n)
        Lg = ops.cholesky(A)
        Lr = scipy.linalg.cholesky(A, lower=True)
        assert np.abs(Lg - Lr).max() <= 1e-11 * np.abs(Lr).max(), n
        assert np.all(np.triu(Lg, 1) == 0)
        for nrhs in (1, 9):
            b = rng.standard_normal((n, nrhs)) if nrhs > 1 else rng.standard_normal(n)
            x = ops.solve_cholesky(Lg, b)
            xr = scipy.linalg.cho_solve((Lr, True), b)
            assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr), (n, nrhs)


def test_random_shapes_gram(oracle):
    from sympgpr_amd import ops
    rng = np.random.default_rng(77)
    for _ in range(12):
        n, n0 = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
        fam = "ABCD"[int(rng.integers(0, 4))]
        x, y = rng.uniform(0, 6.3, n), rng.uniform(-3, 3, n)
        x0, y0 = rng.uniform(0, 6.3, n0), rng.uniform(-3, 3, n0)
        hyp = [0.4, 0.9, 0.6, 1.2] if fam == "D" else [0.4, 0.9, 1.2]
        K = np.full((2 * n, 2 * n0), np.nan, order="F")
        ops.build_k(x, y, x0, y0, hyp, K, family=fam)
        Ko = oracle.build_K(fam, x, y, x0, y0, hyp)
        assert np.all(np.abs(K - Ko) <= 4e-15 * np.abs(Ko).max() + 3e-13 * np.abs(Ko)), (fam, n, n0)


def test_random_sizes_fit(oracle):
    from sympgpr_amd.fit import SympFit
    rng = np.random.default_rng(5)
    for N in sorted(set([1, 2, 63, 64, 65, 255, 257] + list(rng.integers(3, 1400, 10)))):
        fam = "ABCD"[int(rng.integers(0, 4))]
        q, P, z = rng.uniform(0, 6.3, N), rng.uniform(-3, 3, N), rng.standard_normal(2 * N)
        l = 2.0 * np.sqrt(12 * np.pi / max(N, 4))
        hyp = [l, l, 0.5, 1.0] if fam == "D" else [l, l, 1.0]
        s2 = 1e-2 / l**2
        with SympFit(fam, q, P, z, hyp, s2, lower_only=bool(N % 2)) as f:
            a = f.run().alpha()
            nll = f.nll()
        ao, nllo, _ = oracle.fit(fam, q, P, z, hyp, s2)
        assert np.linalg.norm(a - ao) <= 1e-10 * np.linalg.norm(ao), (fam, N)
        assert nll == pytest.approx(nllo, rel=1e-10), (fam, N)


def test_two_threads_two_handles(oracle):
    """different fit handles may be driven from different threads at once (ctypes releases the
    GIL): per-handle workspaces and hand-off words, a thread-local error string and batch arena, the profile
    window behind a mutex, the look-ahead side stream shared per device.  Two threads factor and solve
    different problems concurrently, several times; every result equals the single-threaded oracle fit."""
    import threading
    from sympgpr_amd.fit import SympFit, fit_batch
    rng = np.random.default_rng(77)
    probs = []
    for N in (1536, 2048):
        q, P, z = rng.uniform(0, 2 * np.pi, N), rng.uniform(-3, 3, N), rng.standard_normal(2 * N)
        l = 2.0 * np.sqrt(12 * np.pi / N)
        probs.append((q, P, z, [l, l, 1.0], 1e-2 / l**2))
    want = [oracle.fit("A", *p, threads=4)[:2] for p in probs]
    errs = []

    def work(i):
        try:
            q, P, z, hyp, s2 = probs[i]
            for _ in range(4):
                with SympFit("A", q, P, z, hyp, s2) as f:
                    a, nll = f.run().alpha(), f.nll()
                assert np.linalg.norm(a - want[i][0]) / np.linalg.norm(want[i][0]) < 1e-10
                assert abs(nll - want[i][1]) <= 1e-11 * abs(want[i][1])
                # and a small batched fit from the same thread (per-thread staging arena)
                al, nl, info = fit_batch("A", q[None, :40], P[None, :40], z[None, :80], np.array([hyp]), s2)
                assert info[0] == 0 and np.isfinite(nl[0])
        except Exception as e:      # noqa: BLE001  (re-raised in the main thread)
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]


@pytest.mark.parametrize("n", [640, 768, 896, 1024, 1152, 1280, 1536, 2048, 2560, 3072, 4096, 5120, 6144])
def test_multiples_of_128_factor_and_solve(n):
    """orders the persistent panel kernel (fused chain: solve -> update -> leaf in LDS, strips below, split form) and
    the one-launch strip solves (first strip, one-tile strips, streamed strips) take: L and alpha against LAPACK"""
    from sympgpr_amd import ops
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n + 5))
    A = B @ B.T / n + 0.25 * np.eye(n)
    Lg = ops.cholesky(A)
    Lr = scipy.linalg.cholesky(A, lower=True)
    assert np.abs(Lg - Lr).max() <= 1e-11 * np.abs(Lr).max()
    assert np.all(np.triu(Lg, 1) == 0)
    b = rng.standard_normal(n)
    x = ops.solve_cholesky(Lg, b)
    xr = scipy.linalg.cho_solve((Lr, True), b)
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    assert np.linalg.norm(A @ x - b) <= 1e-11 * np.linalg.norm(b) * np.linalg.cond(A)


@pytest.mark.parametrize("nb,knobs", [(1024, {}), (2048, {}), (1024, {"panel_tile_mult": 3}), (512, {"panel_helpers": 0}),
                                      (1024, {"panel_tiles": 0}), (2048, {"panel_tiles": 0, "panel_below_early": 0})])
def test_wide_panels_with_helpers_and_tiles(nb, knobs):
    """The round-5 pieces of the persistent panel kernel at panel widths the default schedule reaches only at large orders:
    8 and 16 leaf columns per panel = 21 / 105 helper workgroups for the tiles inside the diagonal block (claimed at run time),
    the rows below drawn tile by tile from a counter by more or fewer workgroups than tiles, and each piece switched off in
    turn.  The block width and the tunables are read once per process, so every case is a process of its own
    (tools/potrf_modes.py: factor + solve + residual of Ky alpha = z through the prediction kernel, on n = 4096 and 6144)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SGPR_POTRF_NB=str(nb), SGPR_POTRF_Q="0")
    cmd = [sys.executable, os.path.join(root, "tools", "potrf_modes.py")] + ["%s=%s" % kv for kv in knobs.items()] + ["2048", "3072"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("mode=")]
    assert len(lines) == 2, r.stdout[-1500:]
    for l in lines:
        assert "nb=%d" % nb in l
        assert float(l.split("resid")[1]) < 2e-12, l


def test_three_threads_three_handles_order_16384(oracle):
    """Forward progress with concurrent handles at the size where the panels are wide (n = 16384: 1024- and 512-wide
    panels, up to 31 persistent workgroups each holding a CU): three threads factor three different problems at once,
    twice.  Every panel kernel of the device runs on the one shared side stream (chol.hip, side_stream), so at most one
    set of spinning strips is resident whatever the number of handles; each result must equal the same handle's
    single-threaded result bit for bit (no atomics anywhere on the path) and satisfy Ky alpha = z on sampled rows
    rebuilt by the oracle.  Then the per-device streams are released and come back on demand."""
    import threading
    import torch
    from sympgpr_amd import _lib as L
    from sympgpr_amd.fit import SympFit
    N = 8192
    rng = np.random.default_rng(16384)
    probs = []
    for i in range(3):
        q, P, z = rng.uniform(0, 2 * np.pi, N), rng.uniform(-3, 3, N), rng.standard_normal(2 * N)
        l = 2.0 * np.sqrt(12 * np.pi / N) * (1.0 + 0.1 * i)
        probs.append((q, P, z, [l, l, 1.0], 1e-2 / l**2))
    streams = [torch.cuda.Stream() for _ in probs]        # one stream per handle: the null stream would serialise them
    fits = [SympFit("A", *p, stream=st.cuda_stream) for p, st in zip(probs, streams)]
    try:
        seq = [(f.run().alpha().copy(), f.nll()) for f in fits]
        idx = rng.choice(N, 48, replace=False)
        for (q, P, z, hyp, s2), (a, _) in zip(probs, seq):
            Krows = oracle.build_K("A", q[idx], P[idx], q, P, hyp)          # rows (idx | N + idx) of K
            rows = np.concatenate((idx, N + idx))
            r = Krows @ a + s2 * a[rows] - z[rows]
            assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(z[rows])
        errs = []

        def work(i):
            try:
                for _ in range(2):
                    a, nll = fits[i].run().alpha(), fits[i].nll()
                    assert np.array_equal(a, seq[i][0]), "thread %d: alpha differs from the single-threaded run" % i
                    assert nll == seq[i][1]
            except Exception as e:      # noqa: BLE001  (re-raised in the main thread)
                errs.append(e)
        th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        torch.cuda.synchronize()
        L.check(L.load_library().sgpr_release_device_streams(torch.cuda.current_device()))
        a = fits[0].run().alpha()
        assert np.array_equal(a, seq[0][0])
    finally:
        for f in fits:
            f.close()


def test_queue_factorisation_beside_another_streams_long_kernels(oracle):
    """An order the task-queue Cholesky takes (n = 16384: persistent worker grid + persistent panel kernel on CU-masked
    streams) factored while another thread keeps the same device busy with the long launches of an order-32768
    factorisation on its own stream (look-ahead driver): the persistent grids may be kept off their CUs, switched out and in,
    drained and relaunched -- the results have to be bit-identical to the undisturbed ones."""
    import threading
    import torch
    from sympgpr_amd.fit import SympFit
    rng = np.random.default_rng(4242)

    def problem(N):
        q, P, z = rng.uniform(0, 2 * np.pi, N), rng.uniform(-3, 3, N), rng.standard_normal(2 * N)
        l = 2.0 * np.sqrt(12 * np.pi / N)
        return q, P, z, [l, l, 1.0], 1e-2 / l**2
    pa, pb = problem(8192), problem(16384)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    fa, fb = SympFit("A", *pa, stream=sa.cuda_stream), SympFit("A", *pb, stream=sb.cuda_stream)
    try:
        ref_a = (fa.run().alpha().copy(), fa.nll())
        ref_b = (fb.run().alpha().copy(), fb.nll())
        errs = []

        def work(f, ref, reps):
            try:
                for _ in range(reps):
                    a, nll = f.run().alpha(), f.nll()
                    assert np.array_equal(a, ref[0]) and nll == ref[1]
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(fa, ref_a, 6)), threading.Thread(target=work, args=(fb, ref_b, 2))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        idx = rng.choice(8192, 32, replace=False)
        q, P, z, hyp, s2 = pa
        rows = np.concatenate((idx, 8192 + idx))
        r = oracle.build_K("A", q[idx], P[idx], q, P, hyp) @ ref_a[0] + s2 * ref_a[0][rows] - z[rows]
        assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(z[rows])
    finally:
        fa.close()
        fb.close()

"""world_size > 1 on CPU (gloo): the 2-D block-cyclic driver (sympgpr_amd/dist.py) with the
NumPy block backend must reproduce the single-process oracle fit -- distribution logic,
broadcast / reduce pattern, block bookkeeping, info propagation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, nb, fam, singular, out, serial=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SYMPGPR_NO_TORCH_PRELOAD"] = "1"
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sympgpr_amd.dist import DistFit
        from tests.ref_ops import RefOps
        rng = np.random.default_rng(1234)
        q, P, z = rng.uniform(0, 2 * np.pi, N), rng.uniform(-3, 3, N), rng.standard_normal(2 * N)
        l = 2.0 * np.sqrt(12 * np.pi / N)
        hyp = [l, l, 1.0]
        s2 = 1e-2 / l**2
        if singular:
            # sig < 0 flips the sign of K: Ky is indefinite and a pivot turns negative early
            hyp, s2 = [l, l, -4.0], 2.0 / l**2
        f = DistFit(RefOps(), fam, q, P, z, hyp, s2, nb=nb, serial=serial)
        desc = f.describe(3)
        f.build()
        info = f.factor()
        if info == 0:
            a = f.solve().numpy().copy()
            out[rank] = (0, a, f.nll, desc)
        else:
            out[rank] = (info, None, None)
    finally:
        dist.destroy_process_group()


def _worker_nd(rank, world, port, N, nb, d, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SYMPGPR_NO_TORCH_PRELOAD"] = "1"
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sympgpr_amd.dist import DistFit
        from tests.ref_ops import RefOps
        rng = np.random.default_rng(99)
        X = np.column_stack([rng.uniform(0, 2 * np.pi, (N, d)), rng.uniform(-3, 3, (N, d))])
        z = rng.standard_normal(2 * d * N)
        hyp = np.append(np.full(2 * d, 1.1), 1.0)
        f = DistFit(RefOps(), "A", None, None, z, hyp, 0.05, nb=nb, X=X)
        out[rank] = f.run().numpy().copy()
    finally:
        dist.destroy_process_group()


# the last three: N / nb is NOT a multiple of the grid dimensions -- a rank's coordinate blocks then hold different points
# and the (2d)^2 blocks are written by one call per pair of distinct selections (sgpr_gram_nd_sel_dev)
@pytest.mark.parametrize("world,N,nb,d", [(4, 16, 4, 2), (3, 16, 4, 2), (6, 20, 4, 2), (2, 12, 4, 3), (8, 28, 4, 2)])
def test_block_cyclic_two_pairs_per_point(oracle, world, N, nb, d):
    """BASELINE config 'synthetic d=2 ... 2-D block-cyclic': the distributed driver with d canonical pairs."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_nd, args=(world, _free_port(), N, nb, d, out), nprocs=world, join=True)
    rng = np.random.default_rng(99)
    X = np.column_stack([rng.uniform(0, 2 * np.pi, (N, d)), rng.uniform(-3, 3, (N, d))])
    z = rng.standard_normal(2 * d * N)
    a_o, _, _ = oracle.fit_nd("A", X, z, np.append(np.full(2 * d, 1.1), 1.0), 0.05)
    assert len(out) == world
    for r in range(world):
        assert np.linalg.norm(out[r] - a_o) / np.linalg.norm(a_o) < 1e-11


def _run(world, N, nb, fam="A", singular=False):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), N, nb, fam, singular, out), nprocs=world, join=True)
    return dict(out)


# the last four: N / nb NOT a multiple of the grid dimensions (the q-rows and the P-rows a rank holds
# then belong to different points: the four parts of K are built over their own selections)
@pytest.mark.parametrize("world,N,nb", [(2, 32, 4), (4, 32, 4), (4, 64, 8), (6, 48, 4), (1, 24, 4), (8, 64, 4),
                                        (2, 20, 4), (4, 36, 4), (6, 40, 4), (3, 28, 4)])
def test_block_cyclic_fit_matches_oracle(oracle, world, N, nb):
    res = _run(world, N, nb)
    rng = np.random.default_rng(1234)
    q, P, z = rng.uniform(0, 2 * np.pi, N), rng.uniform(-3, 3, N), rng.standard_normal(2 * N)
    l = 2.0 * np.sqrt(12 * np.pi / N)
    a_o, nll_o, _ = oracle.fit("A", q, P, z, [l, l, 1.0], 1e-2 / l**2)
    assert len(res) == world
    for r in range(world):
        info, a, nll = res[r][:3]
        assert info == 0
        assert np.linalg.norm(a - a_o) / np.linalg.norm(a_o) < 1e-11
        assert nll == pytest.approx(nll_o, rel=1e-11)


def test_block_cyclic_family_c(oracle):
    N, nb = 32, 8
    res = _run(2, N, nb, fam="C")
    rng = np.random.default_rng(1234)
    q, P, z = rng.uniform(0, 2 * np.pi, N), rng.uniform(-3, 3, N), rng.standard_normal(2 * N)
    l = 2.0 * np.sqrt(12 * np.pi / N)
    a_o, _, _ = oracle.fit("C", q, P, z, [l, l, 1.0], 1e-2 / l**2)
    assert np.linalg.norm(res[0][1] - a_o) / np.linalg.norm(a_o) < 1e-11


def test_block_cyclic_not_pd_reports_same_info_everywhere():
    res = _run(4, 32, 4, singular=True)
    infos = {res[r][0] for r in range(4)}
    # same LAPACK-style index as dpotrf on the assembled matrix
    import scipy.linalg
    from oracle.oracle import Oracle
    N = 32
    rng = np.random.default_rng(1234)
    q, P = rng.uniform(0, 2 * np.pi, N), rng.uniform(-3, 3, N)
    l = 2.0 * np.sqrt(12 * np.pi / N)
    Ky = Oracle().build_K("A", q, P, q, P, [l, l, -4.0]) + 2.0 / l**2 * np.eye(2 * N)
    expect = scipy.linalg.lapack.dpotrf(Ky, lower=1)[1]
    assert expect > 0 and infos == {expect}


def _worker_rhs(rank, world, port, N, nb, nrhs, d, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SYMPGPR_NO_TORCH_PRELOAD"] = "1"
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sympgpr_amd.dist import DistFit
        from tests.ref_ops import RefOps
        rng = np.random.default_rng(4321)
        X = np.column_stack([rng.uniform(0, 2 * np.pi, (N, d)), rng.uniform(-3, 3, (N, d))])
        z = rng.standard_normal(2 * d * N)
        B = rng.standard_normal((2 * d * N, nrhs))
        B[:, 0] = z
        hyp = np.append(np.full(2 * d, 1.1), 1.0)
        if d == 1:
            f = DistFit(RefOps(), "A", X[:, 0], X[:, 1], z, hyp[[0, 1, 2]], 0.05, nb=nb)
        else:
            f = DistFit(RefOps(), "A", None, None, z, hyp, 0.05, nb=nb, X=X)
        a = f.run().numpy().copy()
        Xs = f.solve_rhs(B).numpy().copy()
        x1 = f.solve_rhs(B[:, 1]).numpy().copy()              # a single vector goes through the same path
        out[rank] = (a, Xs, x1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,nb,nrhs,d", [(2, 24, 4, 5, 1), (4, 32, 4, 7, 1), (8, 64, 4, 3, 1), (6, 40, 4, 4, 1), (4, 16, 4, 6, 2)])
def test_block_of_right_hand_sides_against_the_distributed_factor(oracle, world, N, nb, nrhs, d):
    """DistFit.solve_rhs: X = Ky^-1 B for a block of right-hand sides kept as rows, one reduce + one broadcast of the whole
    block per block step (SURVEY 8(e) "Solves"; what the reference's matmul(Kyinv, .) of sympgpr.f90:72,85,121 becomes when
    the factor is spread over the grid).  Against a dense solve with the oracle's Ky, on grids with and without N / nb a
    multiple of the grid dimensions; column 0 is z, so it must also equal alpha."""
    out = mp.Manager().dict()
    mp.spawn(_worker_rhs, args=(world, _free_port(), N, nb, nrhs, d, out), nprocs=world, join=True)
    rng = np.random.default_rng(4321)
    X = np.column_stack([rng.uniform(0, 2 * np.pi, (N, d)), rng.uniform(-3, 3, (N, d))])
    z = rng.standard_normal(2 * d * N)
    B = rng.standard_normal((2 * d * N, nrhs))
    B[:, 0] = z
    hyp = np.append(np.full(2 * d, 1.1), 1.0)
    K = oracle.build_K_nd("A", X, X, hyp) if d > 1 else oracle.build_K("A", X[:, 0], X[:, 1], X[:, 0], X[:, 1], hyp[[0, 1, 2]])
    Xr = np.linalg.solve(K + 0.05 * np.eye(K.shape[0]), B)
    assert len(out) == world
    for r in range(world):
        a, Xs, x1 = out[r]
        assert Xs.shape == Xr.shape and x1.shape == (Xr.shape[0], 1)
        assert np.linalg.norm(Xs - Xr) <= 1e-11 * np.linalg.norm(Xr)
        assert np.linalg.norm(Xs[:, 0] - a) <= 1e-12 * np.linalg.norm(a)
        assert np.linalg.norm(x1[:, 0] - Xr[:, 1]) <= 1e-11 * np.linalg.norm(Xr[:, 1])


def test_block_size_is_picked_to_divide():
    from sympgpr_amd.dist import DistFit
    assert DistFit._pick_nb(65536, 2048, 1) == 2048
    assert DistFit._pick_nb(65536, 3000, 1) == 2048
    assert DistFit._pick_nb(12800, 2048, 1) == 1280          # multiple of 128 dividing N
    assert DistFit._pick_nb(12288, 2048, 4) == 1536          # N / nb = 8, a multiple of 4
    assert DistFit._pick_nb(40, 16, 1) == 10
    with pytest.raises(ValueError):
        DistFit._pick_nb(7, 4, 2)


def test_grid_shape():
    from sympgpr_amd.dist import grid_shape
    assert [grid_shape(w) for w in (1, 2, 4, 6, 8)] == [(1, 1), (2, 1), (2, 2), (3, 2), (4, 2)]


def _oracle_fit(family, x, y, z, hyp, sig2n, reg):
    from oracle.oracle import Oracle
    a, nll, _ = Oracle().fit(family, x, y, z, hyp, sig2n)
    return a, nll


def _worker_sections(rank, world, port, nph, Np, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SYMPGPR_NO_TORCH_PRELOAD"] = "1"
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sympgpr_amd import sections
        rng = np.random.default_rng(13)
        xtrain = np.vstack((rng.uniform(0, 2 * np.pi, (Np, nph)), rng.uniform(-3, 3, (Np, nph))))
        ztrain = rng.standard_normal((2 * Np, nph))
        hyp = np.column_stack((rng.uniform(0.4, 0.7, nph), rng.uniform(0.6, 0.9, nph), np.ones(nph)))
        loc = sections.fit_sections("A", xtrain, ztrain, hyp, 1e-3, rank=rank, world=world, fit_fn=_oracle_fit)
        assert sorted(loc) == sections.owned_sections(nph, rank, world)
        alphas, nlls = sections.gather_sections(loc, nph)
        out[rank] = (np.array(alphas), np.array(nlls), xtrain, ztrain, hyp)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nph", [(2, 5), (3, 2)])
def test_sections_are_replicas_only(oracle, world, nph):
    """Split_SympGPR's nphmap independent fits: section m on rank m % world, no data-path
    collective, every rank ends up with every section's weights (also with idle ranks)."""
    Np = 24
    out = mp.Manager().dict()
    mp.spawn(_worker_sections, args=(world, _free_port(), nph, Np, out), nprocs=world, join=True)
    a0, n0, xtrain, ztrain, hyp = out[0]
    for r in range(1, world):
        assert np.array_equal(out[r][0], a0) and np.array_equal(out[r][1], n0)
    for m in range(nph):
        a, nll, _ = oracle.fit("A", xtrain[:Np, m], xtrain[Np:, m], ztrain[:, m], hyp[m], 1e-3)
        assert np.array_equal(a0[m], a) and n0[m] == nll


@pytest.mark.parametrize("world,N,nb", [(2, 24, 4), (4, 48, 8), (8, 64, 8)])
def test_serial_mode_blocking_collectives(oracle, world, N, nb):
    """SGPR_DIST_SERIAL / serial=True: every collective blocking on the compute stream, no side stream, no handles -- the form
    to bisect an RCCL-only failure with.  Same result as the overlapped form; and what every rank says it will do (describe())
    is consistent across the members of each communicator: same roots, same sizes, same order."""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, N, nb, "A", False, out, True), nprocs=world, join=True)
        res = dict(out)
    rng = np.random.default_rng(1234)
    q, P, z = rng.uniform(0, 2 * np.pi, N), rng.uniform(-3, 3, N), rng.standard_normal(2 * N)
    l = 2.0 * np.sqrt(12 * np.pi / N)
    a_o, nll_o, _ = oracle.fit("A", q, P, z, [l, l, 1.0], 1e-2 / l**2)
    for r in range(world):
        info, a, nll, desc = res[r]
        assert info == 0 and np.linalg.norm(a - a_o) <= 1e-10 * np.linalg.norm(a_o) and abs(nll - nll_o) <= 1e-10 * abs(nll_o)
        assert desc["serial"] is True and all(c[3] for st in desc["steps"] for c in st["collectives"])      # all blocking
        assert all(c[4] == "compute" for st in desc["steps"] for c in st["collectives"])
    # per communicator, every member lists the same (root, size) sequence per step
    for r in range(world):
        d = res[r][3]
        for comm in ("row", "col"):
            for peer in d[comm + "_communicator"]:
                dp = res[peer][3]
                for sa, sb in zip(d["steps"], dp["steps"]):
                    mine = [(c[1], c[2]) for c in sa["collectives"] if c[0] == comm]
                    theirs = [(c[1], c[2]) for c in sb["collectives"] if c[0] == comm]
                    assert mine == theirs, (r, peer, comm, sa["K"], mine, theirs)


def test_hbm_plan_of_the_multi_gpu_headline():
    """BASELINE's "synthetic d=2 N=65536" (n = 262144, 550 GB dense) on 2, 4 and 8 MI355X: with the packed lower block
    storage every rank's plan -- local piece of Ky, two sets of panel buffers, diagonal block + workspaces -- fits 288 GB with
    room to spare, ALSO on 2 ranks (SURVEY 8 preamble: "2 GPUs only with lower-triangle storage"); the plans' matrix bytes add
    up to the lower block triangle."""
    from sympgpr_amd.dist import hbm_plan
    N, d, nb = 65536, 2, 2048
    n = 2 * d * N
    nbk = n // nb
    for world, limit in ((8, 60e9), (4, 100e9), (2, 170e9)):
        plans = [hbm_plan(N, d, nb, world, r) for r in range(world)]
        assert sum(p["matrix"] for p in plans) == 8 * nb * nb * nbk * (nbk + 1) // 2
        assert max(p["total"] for p in plans) < limit < 288e9, (world, max(p["total"] for p in plans) / 1e9)
        assert all(p["panel_buffers"] < 0.35 * p["matrix"] for p in plans)
    one = hbm_plan(65536, 1, 2048, 1, 0)
    assert one["matrix"] == 8 * 2048 * 2048 * 64 * 65 // 2

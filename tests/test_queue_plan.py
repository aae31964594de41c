"""The task-queue Cholesky's planner (sympgpr_amd/csrc/cholq.hip, host code): the ordered task list it emits is
replayed on the CPU -- no GPU needed.  Checks: every input of a task is produced by an earlier ticket (or by a panel
kernel whose own inputs are), every tile ends with all its columns applied, and running the list in ticket order on a
dense SPD matrix gives the Cholesky factor (what scipy.linalg.cholesky(lower=True), python/functions/func.py:193 of
the reference, returns)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import queue_sim as qs  # noqa: E402


@pytest.mark.parametrize("n,workers", [(512, 8), (1024, 248), (2048, 248), (2560, 31), (4096, 248), (6144, 240)])
def test_plan_order_and_completeness(n, workers):
    starts, tasks, model_us = qs.fetch_plan(n, workers)
    assert starts[0] == 0 and starts[-1] == n and all(s % 256 == 0 for s in starts)
    assert qs.check_order(n, starts, tasks) == len(tasks)
    assert model_us > 0


@pytest.mark.parametrize("n,workers", [(768, 8), (1536, 8), (2560, 248)])
def test_plan_replay_is_the_cholesky_factor(n, workers):
    starts, tasks, _ = qs.fetch_plan(n, workers)
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    A = B @ B.T / n + np.eye(n)
    L = qs.replay(A, starts, tasks)
    Lr = np.linalg.cholesky(A)
    assert np.abs(L - Lr).max() <= 1e-13 * np.abs(Lr).max()


def test_plan_update_ranges_grow_away_from_the_diagonal():
    """far tiles pile up several panels per update (that is the point of the planner): at n = 16384 most of the
    flop is in updates spanning more than one 512-wide panel, and none spans more than 2048 columns"""
    starts, tasks, _ = qs.fetch_plan(16384, 248)
    ks = np.array([qs.unpack(t)[5] - qs.unpack(t)[4] for t in tasks if qs.unpack(t)[0] == qs.TASK_U])
    assert ks.min() >= 1 and ks.max() <= 16
    assert ks[ks > 4].sum() > 0.5 * ks.sum()


def test_plan_model_replay_matches_the_planner():
    """the in-order replay on the planner's own cost model ends when the planner said it would"""
    n, workers = 4096, 248
    starts, tasks, model_us = qs.fetch_plan(n, workers)
    end, busy, wait = qs.simulate(n, starts, tasks, workers)
    assert abs(end - model_us) <= 0.02 * model_us + 5.0


@pytest.mark.parametrize("n,workers,nq", [(2048, 248, 1), (2560, 31, 3), (4096, 248, 4), (6144, 240, 11), (16384, 248, -1)])
def test_partial_plan_hands_over_a_fully_updated_block(n, workers, nq):
    """the queue factors the first nq panels only (SGPR_Q_TAIL; by default it runs them all): every tile of the
    block it leaves carries all updates of those panels, every row strip below them is solved, no task touches a later panel"""
    starts, tasks, model_us, used = qs.fetch_plan_partial(n, workers, nq)
    assert used == (nq if nq > 0 else used) and 1 <= used <= len(starts) - 1
    if nq < 0:
        assert used == len(starts) - 1          # the default hand-over point: none (SGPR_Q_TAIL=0), the queue runs every panel
    assert qs.check_order(n, starts, tasks, used) == len(tasks)
    for t in tasks:
        typ, k, i, j, a, b = qs.unpack(t)
        assert (k < used) if typ == qs.TASK_T else (128 * b <= starts[used])


@pytest.mark.parametrize("n,workers,nq", [(1536, 8, 1), (2560, 248, 2), (2560, 8, 4)])
def test_partial_plan_replay_then_dense_tail_is_the_cholesky_factor(n, workers, nq):
    starts, tasks, _, used = qs.fetch_plan_partial(n, workers, nq)
    rng = np.random.default_rng(n + nq)
    B = rng.standard_normal((n, n))
    A = B @ B.T / n + np.eye(n)
    L = qs.replay(A, starts, tasks, used)
    Lr = np.linalg.cholesky(A)
    assert np.abs(L - Lr).max() <= 1e-13 * np.abs(Lr).max()

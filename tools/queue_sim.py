"""The task-queue Cholesky's plan (csrc/cholq.h) on the CPU: fetch the ordered task list from the library (host code,
no GPU needed), check its ordering invariant, replay it numerically with NumPy, and run it through a discrete-event
model of the worker grid + the chain of diagonal blocks to see where workers would wait.
`python tools/queue_sim.py N [workers]` prints the plan's summary.  Used by tests/test_queue_plan.py."""
import ctypes as C
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TM, TN, LEAF = 256, 128, 128
TASK_U, TASK_T = 0, 1


def fetch_plan(n, nworkers):
    """-> (starts, tasks[ntasks, 2] uint32, the planner's own time estimate in us): the whole factorisation in the queue"""
    from sympgpr_amd import _lib as L
    probe = L.load_probe_library()
    counts = (C.c_int * 3)()
    L.check(probe.sgpr_probe_queue_plan(n, nworkers, None, 0, None, 0, counts), "sgpr_probe_queue_plan")
    nblk, ntasks = counts[0], counts[1]
    starts = (C.c_int * (nblk + 1))()
    tasks = (C.c_uint * max(2 * ntasks, 2))()
    L.check(probe.sgpr_probe_queue_plan(n, nworkers, starts, nblk + 1, tasks, ntasks, counts), "sgpr_probe_queue_plan")
    t = np.frombuffer(tasks, dtype=np.uint32, count=2 * ntasks).reshape(-1, 2).copy()
    return list(starts), t, float(counts[2])


def fetch_plan_partial(n, nworkers, nq=-1):
    """-> (starts, tasks, model us, nq): the queue factors panels 0 .. nq-1 and hands the rest to the look-ahead driver
    (nq < 0: the default hand-over point of this order)"""
    from sympgpr_amd import _lib as L
    probe = L.load_probe_library()
    counts = (C.c_int * 4)()
    L.check(probe.sgpr_probe_queue_plan_partial(n, nworkers, nq, None, 0, None, 0, counts), "sgpr_probe_queue_plan_partial")
    nblk, ntasks = counts[0], counts[1]
    starts = (C.c_int * (nblk + 1))()
    tasks = (C.c_uint * max(2 * ntasks, 2))()
    L.check(probe.sgpr_probe_queue_plan_partial(n, nworkers, nq, starts, nblk + 1, tasks, ntasks, counts),
            "sgpr_probe_queue_plan_partial")
    t = np.frombuffer(tasks, dtype=np.uint32, count=2 * ntasks).reshape(-1, 2).copy()
    return list(starts), t, float(counts[2]), int(counts[3])


def unpack(t):
    """-> (type, k, i, j, a, b)"""
    w0, w1 = int(t[0]), int(t[1])
    return w0 >> 30, (w0 >> 21) & 511, (w0 >> 11) & 1023, w0 & 2047, w1 >> 16, w1 & 0xFFFF


def chain_tiles(k, starts):
    """tiles the diagonal strips of panel k wait for"""
    j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
    return sorted({(starts[k] // TM + g // 2, j0 + c) for g in range(W) for c in range(g + 1)})


def check_order(n, starts, tasks, nq=None):
    """Every input of a task is produced by a task with a smaller ticket, or by a panel kernel whose own inputs
    are; every tile ends up with all its columns applied, every row strip solved against every panel left of it.
    The panel kernels are run as late as possible here: a task that needs one gets it only if ITS inputs are there.
    nq: the queue factors panels 0 .. nq-1 only; what is checked then is the state it hands over -- every tile of the
    block that is left carries all updates of those panels, every row strip below them is solved against all of them."""
    tm, tn = n // TM, n // TN
    nblk = len(starts) - 1
    nq = nblk if nq is None else nq
    S = starts[nq]
    ver = np.zeros((tm, tn), dtype=np.int64)
    tver = np.zeros(tn, dtype=np.int64)
    chain_done = set()

    def run_chain(k):
        """panel kernel k: diagonal strips, and the rows of the next diagonal block (all their tiles there first)"""
        assert k == 0 or (k - 1) in chain_done, ("chain order", k)
        assert k < nq, ("a panel beyond the queue's part", k)
        need = starts[k] // LEAF
        for (vi, vj) in chain_tiles(k, starts):
            assert ver[vi, vj] >= need, ("chain", k, vi, vj, ver[vi, vj])
        if k + 1 < nblk:
            j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
            for r in range(starts[k + 1] // LEAF, starts[k + 2] // LEAF):
                for c in range(W):
                    assert ver[r // 2, j0 + c] >= need, ("band", k, r, c)
                assert tver[r] == need
                tver[r] = starts[k + 1] // LEAF
        chain_done.add(k)

    def need_chain(k):
        for q in range(len(chain_done), k + 1):
            run_chain(q)

    def need_tver(r, v):
        """row strip r final through column block v: if a panel kernel's band produces it, that kernel must be runnable"""
        if tver[r] >= v:
            return
        k = max(q for q in range(nblk) if starts[q] // LEAF < v)           # the panel whose columns end the range
        assert k + 1 < nblk and starts[k + 1] // LEAF <= r < starts[k + 2] // LEAF, ("tver", r, v, tver[r])
        need_chain(k)
        assert tver[r] >= v, ("tver after panel kernel", r, v, tver[r])

    for t in tasks:
        typ, k, i, j, a, b = unpack(t)
        if typ == TASK_U:
            assert 256 * i + 255 >= 128 * j and a < b
            assert ver[i, j] == a, ("ver", (i, j, a, b), ver[i, j])
            for r in (2 * i, 2 * i + 1, j):
                need_tver(r, b)
            capj = min(max(s for s in starts if s <= 128 * j), S) // LEAF
            assert b <= capj
            ver[i, j] = b
        else:
            need_chain(k)
            base = starts[k] // LEAF
            j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
            assert k + 2 < len(starts) and 256 * i >= starts[k + 2]
            for c in range(W):
                assert ver[i, j0 + c] >= base, ("solve", k, i, c)
            assert tver[2 * i] == base and tver[2 * i + 1] == base, ("solve order", k, i, tver[2 * i])
            tver[2 * i] = tver[2 * i + 1] = starts[k + 1] // LEAF
    need_chain(nq - 1)
    for i in range(tm):
        for j in range(tn):
            if 256 * i + 255 >= 128 * j:
                capj = min(max(s for s in starts if s <= 128 * j), S) // LEAF
                assert ver[i, j] == capj, (i, j, ver[i, j], capj)
    for k in range(min(nq, nblk - 1)):
        for r in range(starts[k + 1] // LEAF, tn):
            assert tver[r] >= starts[k + 1] // LEAF
    return len(tasks)


def replay(A, starts, tasks, nq=None):
    """Run the list in ticket order on a dense SPD matrix (lower triangle significant); a panel kernel (diagonal
    block + the rows of the next one) runs as soon as its inputs are complete.  Returns L (lower).
    nq: the queue's part ends behind panel nq - 1; the block that is left is then factored in one piece (the look-ahead
    driver's job on the device)."""
    import scipy.linalg
    A = np.array(A, dtype=np.float64, order="F")
    n = A.shape[0]
    tm, tn = n // TM, n // TN
    nblk = len(starts) - 1
    nq = nblk if nq is None else nq
    ver = np.zeros((tm, tn), dtype=np.int64)
    done = []

    def chains():
        k = len(done)
        while k < nq:
            need = starts[k] // LEAF
            tiles = set(chain_tiles(k, starts))
            if k + 1 < nblk:
                j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
                tiles |= {(r // 2, j0 + c) for r in range(starts[k + 1] // LEAF, starts[k + 2] // LEAF) for c in range(W)}
            if any(ver[t_] < need for t_ in tiles):
                return
            s0, s1 = starts[k], starts[k + 1]
            D = np.tril(A[s0:s1, s0:s1])
            A[s0:s1, s0:s1] = np.linalg.cholesky(D + np.tril(D, -1).T)
            if k + 1 < nblk:
                r = slice(starts[k + 1], starts[k + 2])
                A[r, s0:s1] = scipy.linalg.solve_triangular(np.tril(A[s0:s1, s0:s1]), A[r, s0:s1].T, lower=True).T
            done.append(k)
            k += 1

    chains()
    for t in tasks:
        typ, k, i, j, a, b = unpack(t)
        if typ == TASK_T:
            assert k in done
            s0, s1 = starts[k], starts[k + 1]
            r = slice(TM * i, TM * i + TM)
            A[r, s0:s1] = scipy.linalg.solve_triangular(np.tril(A[s0:s1, s0:s1]), A[r, s0:s1].T, lower=True).T
        else:
            r, c, kk = slice(TM * i, TM * i + TM), slice(TN * j, TN * j + TN), slice(LEAF * a, LEAF * b)
            A[r, c] -= A[r, kk] @ A[c, kk].T
            ver[i, j] = b
        chains()
    assert len(done) == nq
    if nq < nblk:
        S = starts[nq]
        D = np.tril(A[S:, S:])
        A[S:, S:] = np.linalg.cholesky(D + np.tril(D, -1).T)
    return np.tril(A)


def simulate(n, starts, tasks, nworkers, leaf_us=72.0, pair_us=17.0, kstep_us=3.56, fixed_us=28.0, band0=40.0, band_pair=12.0):
    """The in-order execution of the list on `nworkers` workers (each takes the next ticket when free and waits for
    the task's inputs) beside the chain, on the planner's cost model.  Returns (makespan_us, busy_us, wait_us)."""
    tm, tn = n // TM, n // TN
    nblk = len(starts) - 1
    ver_t = {(i, j, 0): 0.0 for i in range(tm) for j in range(tn)}
    tver_t = {(r, 0): 0.0 for r in range(tn)}
    chain_t = {}

    def chain(k):
        if k in chain_t:
            return chain_t[k]
        W = (starts[k + 1] - starts[k]) // LEAF
        need = starts[k] // LEAF
        t0 = 0.0 if k == 0 else chain(k - 1)
        t_in = max([ver_t.get((vi, vj, need), np.inf) for (vi, vj) in chain_tiles(k, starts)] + [t0])
        chain_t[k] = t_in + leaf_us * W + pair_us * W * (W - 1) / 2
        if k + 1 < nblk:
            j0 = starts[k] // TN
            lag = band0 + band_pair * W * (W - 1) / 2
            rows = range(starts[k + 1] // LEAF, starts[k + 2] // LEAF)
            rin = max(ver_t.get((r // 2, j0 + c, need), np.inf) for r in rows for c in range(W))
            for r in rows:
                tver_t[(r, starts[k + 1] // LEAF)] = max(chain_t[k], rin) + lag
        return chain_t[k]

    def tver_time(r, v):
        if (r, v) not in tver_t:
            chain(max(q for q in range(nblk) if starts[q] // LEAF < v))
        return tver_t.get((r, v), np.inf)

    free = [(0.0, wk) for wk in range(nworkers)]
    heapq.heapify(free)
    busy = wait = end = 0.0
    for t in tasks:
        tw, wk = heapq.heappop(free)
        typ, k, i, j, a, b = unpack(t)
        if typ == TASK_U:
            ready = max(ver_t.get((i, j, a), np.inf), tver_time(2 * i, b), tver_time(2 * i + 1, b), tver_time(j, b))
            c_us = kstep_us * 8 * (b - a) + fixed_us
        else:
            base = starts[k] // LEAF
            j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
            ready = max([ver_t.get((i, j0 + c, base), np.inf) for c in range(W)] + [chain(k)])
            c_us = kstep_us * (sum(8 * c for c in range(1, W)) + 8 * W) + (2 * W - 1) * fixed_us * 0.5
        assert np.isfinite(ready), unpack(t)
        start = max(tw, ready)
        fin = start + c_us
        wait += max(0.0, ready - tw)
        busy += c_us
        if typ == TASK_U:
            ver_t[(i, j, b)] = fin
        else:
            tver_t[(2 * i, starts[k + 1] // LEAF)] = tver_t[(2 * i + 1, starts[k + 1] // LEAF)] = fin
        end = max(end, fin)
        heapq.heappush(free, (fin, wk))
    end = max(end, chain(nblk - 1))
    return end, busy, wait


if __name__ == "__main__":
    n = int(sys.argv[1])
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 248
    starts, tasks, model_us = fetch_plan(n, nw)
    ks = np.array([(unpack(t)[5] - unpack(t)[4]) for t in tasks if unpack(t)[0] == TASK_U])
    print("n=%d panels=%d (width %d) tasks=%d  mean k of an update %.0f  planner estimate %.2f ms (%.1f TFLOP/s)" % (
        n, len(starts) - 1, starts[1], len(tasks), 128 * ks.mean(), model_us / 1e3, n**3 / 3 / max(model_us, 1) / 1e6))
    print("k histogram (x128):", {int(k): int(c) for k, c in enumerate(np.bincount(ks)) if c})
    if n <= 8192:
        check_order(n, starts, tasks)
    end, busy, wait = simulate(n, starts, tasks, nw)
    print("in-order replay on the model: %.2f ms  workers busy %.1f %%  waiting %.1f %%" % (end / 1e3, 100 * busy / (end * nw), 100 * wait / (end * nw)))

"""The task-queue Cholesky's plan (csrc/cholq.h) on the CPU: fetch the ordered task list from the library (host code,
no GPU needed), check its ordering invariant, replay it numerically with NumPy, and run it through a discrete-event
model of the worker grid + the chain of diagonal blocks (costs from the measured kernel constants in DESIGN.md) to
see where workers would wait.  `python tools/queue_sim.py N [workers]` prints the model's timeline summary.

Used by tests/test_queue_plan.py; also the offline tuning aid for the panel schedule (SGPR_Q_W0/T0/T1/T2/WMAX)."""
import ctypes as C
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TM, TN, LEAF = 256, 128, 128
TASK_U, TASK_T = 0, 1


def fetch_plan(n, nworkers):
    from sympgpr_amd import _lib as L
    probe = L.load_probe_library()
    counts = (C.c_int * 2)()
    L.check(probe.sgpr_probe_queue_plan(n, nworkers, None, 0, None, 0, counts), "sgpr_probe_queue_plan")
    nblk, ntasks = counts[0], counts[1]
    starts = (C.c_int * (nblk + 1))()
    tasks = (C.c_uint * max(ntasks, 1))()
    L.check(probe.sgpr_probe_queue_plan(n, nworkers, starts, nblk + 1, tasks, ntasks, counts), "sgpr_probe_queue_plan")
    t = np.frombuffer(tasks, dtype=np.uint32, count=ntasks).copy()
    return list(starts), t


def unpack(t):
    return int(t >> 30), int((t >> 21) & 511), int((t >> 11) & 1023), int(t & 2047)


def deps_of(task, starts):
    """What a task waits for: list of ('ver', i, j, need) / ('tver', i, need) / ('chain', k)."""
    typ, k, i, j = unpack(task)
    if typ == TASK_U:
        return [("ver", i, j, k), ("tver", 2 * i, k + 1), ("tver", 2 * i + 1, k + 1), ("tver", j, k + 1)]
    j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
    return [("ver", i, j0 + c, k) for c in range(W)] + [("chain", k)]


def chain_deps(k, starts):
    """tiles the diagonal strips of panel k wait for (all at version k)"""
    j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
    out = []
    for g in range(W):
        vi = starts[k] // TM + g // 2
        out += [(vi, j0 + c) for c in range(g + 1)]
    return sorted(set(out))


def check_order(n, starts, tasks):
    """Every dependency of a task is produced by a task with a smaller ticket (or by the chain, whose own
    inputs come from smaller tickets than its first consumer).  Returns the number of tasks checked."""
    tm, tn = n // TM, n // TN
    ver = np.zeros((tm, tn), dtype=np.int64)
    tver = np.zeros(tn, dtype=np.int64)      # per 128-row strip
    chain_done = set()
    nblk = len(starts) - 1
    seen = set()

    def run_chain(k):
        """the panel kernel of panel k: its diagonal strips, and the rows of the next diagonal block beside them"""
        for (vi, vj) in chain_deps(k, starts):
            assert ver[vi, vj] == k, ("chain", k, vi, vj, ver[vi, vj])
        if k + 1 < nblk:
            j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
            for r in range(starts[k + 1] // LEAF, starts[k + 2] // LEAF):
                for c in range(W):
                    assert ver[r // 2, j0 + c] == k, ("band", k, r, c)
                assert tver[r] == k
                tver[r] = k + 1
        chain_done.add(k)

    for t in tasks:
        typ, k, i, j = unpack(t)
        assert t not in seen, "duplicate task"
        seen.add(int(t))
        for d in deps_of(t, starts):
            if d[0] == "ver":
                assert ver[d[1], d[2]] == d[3], ("ver", unpack(t), d, ver[d[1], d[2]])
            elif d[0] == "tver":
                pass        # checked below, once the panel kernel that may produce it has been given its chance
            else:
                if d[1] not in chain_done:
                    run_chain(d[1])
        if typ == TASK_U:
            # inputs solved by the panel kernel itself (rows of the next diagonal block): run it when first asked for
            if k not in chain_done and all(ver[vi, vj] == k for (vi, vj) in chain_deps(k, starts)):
                run_chain(k)
            for d in deps_of(t, starts):
                if d[0] == "tver":
                    assert tver[d[1]] >= d[2], ("tver after chain", unpack(t), d, tver[d[1]])
            assert 256 * i + 255 >= 128 * j and 128 * j >= starts[k + 1] and 256 * i >= starts[k + 1]
            ver[i, j] = k + 1
        else:
            assert 256 * i >= starts[k + 2]
            assert tver[2 * i] == k and tver[2 * i + 1] == k
            tver[2 * i] = tver[2 * i + 1] = k + 1
    # completeness: every lower tile right of panel k got panel k's update, every row tile below every panel was solved
    for k in range(nblk - 1):
        for r in range(starts[k + 1] // LEAF, tn):
            assert tver[r] >= k + 1, (k, r, tver[r])
    for i in range(tm):
        for j in range(tn):
            if 256 * i + 255 >= 128 * j:
                kk = max(q for q in range(nblk) if starts[q] <= 128 * j)      # panel that holds column tile j
                assert ver[i, j] == kk, (i, j, ver[i, j], kk)
    return len(tasks)


def replay(A, starts, tasks):
    """Run the list in ticket order on a dense SPD matrix (lower triangle significant); the chain of a panel runs
    when its first consumer asks for it.  Returns L (lower)."""
    import scipy.linalg
    A = np.array(A, dtype=np.float64, order="F")
    done = set()

    def chain(k):
        s0, s1 = starts[k], starts[k + 1]
        A[s0:s1, s0:s1] = np.linalg.cholesky(np.tril(A[s0:s1, s0:s1]) + np.tril(A[s0:s1, s0:s1], -1).T)
        if k + 2 < len(starts):
            r = slice(starts[k + 1], starts[k + 2])
            A[r, s0:s1] = scipy.linalg.solve_triangular(np.tril(A[s0:s1, s0:s1]), A[r, s0:s1].T, lower=True).T
        done.add(k)

    for t in tasks:
        typ, k, i, j = unpack(t)
        s0, s1 = starts[k], starts[k + 1]
        if typ == TASK_T:
            if k not in done:
                chain(k)
            Lkk = np.tril(A[s0:s1, s0:s1])
            r = slice(TM * i, TM * i + TM)
            A[r, s0:s1] = scipy.linalg.solve_triangular(Lkk, A[r, s0:s1].T, lower=True).T
        else:
            if k not in done:
                chain(k)
            r, c = slice(TM * i, TM * i + TM), slice(TN * j, TN * j + TN)
            A[r, c] -= A[r, s0:s1] @ A[c, s0:s1].T
    last = len(starts) - 2
    if last not in done:
        chain(last)
    return np.tril(A)


def simulate(n, starts, tasks, nworkers, leaf_us=72.0, kstep_us=3.56, fixed_us=18.0, sync_us=3.0):
    """Discrete-event model: workers draw tickets in order and hold two (the running one and the next);
    a task starts when its inputs are there.  The chain of panel k starts when its tiles are there and takes
    leaf_us per leaf column.  Returns (makespan_us, busy_us, wait_us, chain_spans)."""
    tm, tn = n // TM, n // TN
    nblk = len(starts) - 1
    ver_t = {}      # (i, j, version) -> time that version was published
    tver_t = {}     # (i, version)
    for i in range(tm):
        for j in range(tn):
            ver_t[(i, j, 0)] = 0.0
    for r in range(tn):
        tver_t[(r, 0)] = 0.0
    chain_t = {}
    chain_span = {}

    def chain_ready(k):
        if k in chain_t:
            return chain_t[k]
        W = (starts[k + 1] - starts[k]) // LEAF
        t0 = 0.0 if k == 0 else chain_t.get(k - 1, 0.0)
        t_in = max([ver_t.get((vi, vj, k), np.inf) for (vi, vj) in chain_deps(k, starts)] + [t0])
        chain_span[k] = (t_in, t_in + leaf_us * W)
        chain_t[k] = t_in + leaf_us * W
        if k + 1 < nblk:
            # the rows of the next diagonal block, solved beside the chain by workgroups of the panel kernel
            j0 = starts[k] // TN
            work = (W + W * (W - 1) / 2) * 20.0
            for r in range(starts[k + 1] // LEAF, starts[k + 2] // LEAF):
                rin = max(ver_t.get((r // 2, j0 + c, k), np.inf) for c in range(W))
                tver_t[(r, k + 1)] = max(chain_t[k] + 40.0, rin + work)
        return chain_t[k]

    def cost(task):
        typ, k, i, j = unpack(task)
        w = starts[k + 1] - starts[k]
        if typ == TASK_U:
            return kstep_us * w / 16 + fixed_us
        W = w // LEAF
        ksteps = sum(8 * c for c in range(1, W)) + 8 * W
        return kstep_us * ksteps + (2 * W - 1) * 12.0

    free = [(0.0, wk) for wk in range(nworkers)]
    heapq.heapify(free)
    busy = wait = 0.0
    end = 0.0
    for t in tasks:
        tw, wk = heapq.heappop(free)
        typ, k, i, j = unpack(t)
        ready = 0.0
        for d in deps_of(t, starts):
            if d[0] == "ver":
                ready = max(ready, ver_t.get((d[1], d[2], d[3]), np.inf))
            elif d[0] == "tver":
                if (d[1], d[2]) not in tver_t:
                    chain_ready(k)
                ready = max(ready, tver_t.get((d[1], d[2]), np.inf))
            else:
                ready = max(ready, chain_ready(d[1]))
        assert np.isfinite(ready), unpack(t)
        start = max(tw, ready) + sync_us
        c = cost(t)
        fin = start + c
        wait += max(0.0, ready - tw)
        busy += c
        if typ == TASK_U:
            ver_t[(i, j, k + 1)] = fin
        else:
            tver_t[(2 * i, k + 1)] = tver_t[(2 * i + 1, k + 1)] = fin
        end = max(end, fin)
        heapq.heappush(free, (fin, wk))
    end = max(end, chain_ready(nblk - 1))
    return end, busy, wait, chain_span


if __name__ == "__main__":
    n = int(sys.argv[1])
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 248
    starts, tasks = fetch_plan(n, nw)
    print("n=%d panels=%d widths=%s tasks=%d" % (n, len(starts) - 1, np.diff(starts).tolist(), len(tasks)))
    check_order(n, starts, tasks)
    end, busy, wait, spans = simulate(n, starts, tasks, nw)
    print("model: %.2f ms  (= %.1f TFLOP/s)  workers busy %.1f %%  waiting %.1f %%" % (
        end / 1e3, n**3 / 3 / end / 1e6, 100 * busy / (end * nw), 100 * wait / (end * nw)))

"""Prototype of a simulation-driven planner for the task-queue Cholesky: variable-k update tasks U(i, j, a, b)
(tile (i, j) -= L(i, a:b) L(j, a:b)^T in units of 128 columns), chosen greedily by a list scheduler that always
gives the most urgent tile everything that is available for it.  python tools/queue_plan_proto.py n [workers]"""
import heapq
import sys

import numpy as np

TM, TN, LEAF = 256, 128, 128


def plan(n, starts, nworkers, leaf_us=72.0, kstep_us=3.56, fixed_us=25.0, band_lag_us=40.0, kcap=16, verbose=False):
    tm, tn = n // TM, n // TN
    nblk = len(starts) - 1
    pcol = np.zeros(tn, dtype=int)          # panel of column tile j
    for k in range(nblk):
        pcol[starts[k] // TN:starts[k + 1] // TN] = k
    # state
    ver = np.zeros((tm, tn), dtype=int)      # leaf columns applied to tile (i, j)
    busy = np.zeros((tm, tn), dtype=bool)
    rowfin = np.zeros(tn, dtype=int)         # leading leaf columns of strip r that are final
    rowfin_t = {}                            # (r, value) -> time
    chain_end = {}
    tsolved = np.zeros((tm,), dtype=int)     # panels solved for row tile i (T tasks), as count
    events = []                              # (time, kind, payload)
    tasks = []                               # emitted in start order
    lower = lambda i, j: TM * i + TM - 1 >= TN * j
    need_final = lambda j: starts[pcol[j]] // LEAF       # columns a tile in column tile j must have applied before its panel touches it

    # chain bookkeeping: chain k can start when the diag-block tiles are fully updated and chain k-1 has ended
    chain_started = set()

    def chain_tiles(k):
        j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
        out = set()
        for g in range(W):
            vi = starts[k] // TM + g // 2
            for c in range(g + 1):
                out.add((vi, j0 + c))
        return out

    def band_tiles(k):
        if k + 1 >= nblk:
            return set()
        j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
        return {(r // 2, j0 + c) for r in range(starts[k + 1] // LEAF, starts[k + 2] // LEAF) for c in range(W)}

    now = 0.0
    free = [(0.0, w) for w in range(nworkers)]
    heapq.heapify(free)
    pending_chain = 0                       # next chain to start
    t_pending = {}                          # panel -> list of row tiles whose T is not yet issued
    for k in range(nblk):
        first = starts[k + 2] // TM if k + 2 <= nblk else tm
        t_pending[k] = list(range(first, tm))

    def try_start_chain(t):
        nonlocal pending_chain
        while pending_chain < nblk:
            k = pending_chain
            need = starts[k] // LEAF
            if any(ver[i, j] < need or busy[i, j] for (i, j) in chain_tiles(k)):
                return
            t0 = max(t, chain_end.get(k - 1, 0.0))
            W = (starts[k + 1] - starts[k]) // LEAF
            te = t0 + leaf_us * W
            chain_end[k] = te
            heapq.heappush(events, (te, 0, ("chain", k)))
            # band rows: need their own tiles
            pending_chain += 1
            if pending_chain < nblk and te > t:
                return  # next chain cannot start before this one ends anyway

    band_wait = {}     # k -> set of band tiles not yet at need

    def finish_chain(k, t):
        # diag strips final
        s1 = starts[k + 1] // LEAF
        for r in range(starts[k] // LEAF, s1):
            rowfin[r] = s1
        # band strips: final band_lag after chain end (assuming their tiles were ready)
        if k + 1 < nblk:
            heapq.heappush(events, (t + band_lag_us, 0, ("band", k)))

    def finish_band(k, t):
        s1 = starts[k + 1] // LEAF
        need = starts[k] // LEAF
        bt = band_tiles(k)
        if any(ver[i, j] < need or busy[i, j] for (i, j) in bt):
            heapq.heappush(events, (t + 20.0, 0, ("band", k)))      # poll again (tiles late)
            return
        for r in range(starts[k + 1] // LEAF, starts[k + 2] // LEAF):
            rowfin[r] = s1

    def pick(t):
        """most urgent ready task at time t, or None"""
        best = None
        # T tasks
        for k in range(nblk):
            if k not in chain_end or chain_end[k] > t or not t_pending[k]:
                continue
            need = starts[k] // LEAF
            j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
            for i in t_pending[k]:
                if all(ver[i, j0 + c] >= need and not busy[i, j0 + c] for c in range(W)):
                    key = (k, 0, i)
                    if best is None or key < best[0]:
                        best = (key, ("T", k, i))
                    break
            if best is not None and best[0][0] == k:
                break
        # U tasks: scan column tiles in order; the first panels are the urgent ones
        for j in range(tn):
            p = pcol[j]
            if best is not None and best[0][0] < p:
                break
            cap = need_final(j)
            found = False
            for i in range(j // 2, tm):
                if not lower(i, j) or busy[i, j]:
                    continue
                a = ver[i, j]
                if a >= cap:
                    continue
                b = min(rowfin[2 * i], rowfin[2 * i + 1], rowfin[j], cap, a + kcap)
                if b > a:
                    key = (p, 1, i)
                    if best is None or key < best[0]:
                        best = (key, ("U", i, j, a, b))
                    found = True
                    break
            if found and best[0][0] == p:
                break
        return None if best is None else best[1]

    def cost(task):
        if task[0] == "U":
            return kstep_us * 8 * (task[4] - task[3]) + fixed_us
        k = task[1]
        W = (starts[k + 1] - starts[k]) // LEAF
        ksteps = sum(8 * c for c in range(1, W)) + 8 * W
        return kstep_us * ksteps + (2 * W - 1) * 12.0

    try_start_chain(0.0)
    busy_us = 0.0
    end = 0.0
    idle_workers = []
    while True:
        # process all events up to `now`
        while events and events[0][0] <= now:
            te, _, ev = heapq.heappop(events)
            if ev[0] == "chain":
                finish_chain(ev[1], te)
            elif ev[0] == "band":
                finish_band(ev[1], te)
            elif ev[0] == "done":
                task = ev[1]
                if task[0] == "U":
                    _, i, j, a, b = task
                    ver[i, j] = b
                    busy[i, j] = False
                else:
                    _, k, i = task
                    s1 = starts[k + 1] // LEAF
                    rowfin[2 * i] = rowfin[2 * i + 1] = s1
                    j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
                    for c in range(W):
                        busy[i, j0 + c] = False
            try_start_chain(te)
        # hand out work to free workers
        progressed = False
        while free and free[0][0] <= now:
            task = pick(now)
            if task is None:
                break
            tw, w = heapq.heappop(free)
            c = cost(task)
            if task[0] == "U":
                busy[task[1], task[2]] = True
            else:
                k, i = task[1], task[2]
                t_pending[k].remove(i)
                j0, W = starts[k] // TN, (starts[k + 1] - starts[k]) // LEAF
                for cc in range(W):
                    busy[i, j0 + cc] = True
            tasks.append((now, task))
            busy_us += c
            heapq.heappush(events, (now + c, 0, ("done", task)))
            heapq.heappush(free, (now + c, w))
            end = max(end, now + c)
            progressed = True
        # advance time
        nxt = []
        if events:
            nxt.append(events[0][0])
        if free and free[0][0] > now:
            nxt.append(free[0][0])
        if not nxt:
            break
        tn_ = min(nxt)
        if tn_ <= now and not progressed:
            tn_ = now + 1.0
        now = max(now, tn_)
        if pending_chain >= nblk and not events and all(not v for v in t_pending.values()):
            if all(ver[i, j] >= need_final(j) for i in range(tm) for j in range(tn) if lower(i, j)):
                break
    end = max(end, max(chain_end.values()))
    return end, busy_us, tasks, chain_end


if __name__ == "__main__":
    n = int(sys.argv[1])
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 240
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    starts = list(range(0, n, w)) + [n]
    end, busy, tasks, chain_end = plan(n, starts, nw)
    ks = np.array([t[1][4] - t[1][3] for t in tasks if t[1][0] == "U"])
    print("n=%d panels of %d: %.2f ms (%.1f TFLOP/s), busy %.1f %%, %d tasks, mean k = %.0f, chain ends %.2f ms" % (
        n, w, end / 1e3, n**3 / 3 / end / 1e6, 100 * busy / (end * nw), len(tasks), 128 * ks.mean(), max(chain_end.values()) / 1e3))
    hist = np.bincount(ks)
    print("k histogram (x128):", {int(k): int(c) for k, c in enumerate(hist) if c})

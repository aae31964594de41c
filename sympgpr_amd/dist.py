"""2-D block-cyclic Gram build + Cholesky + alpha across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  The matrix Ky (order n = 2N) is cut into nb x nb blocks; block (I, J) lives
on grid position (I mod pr, J mod pc).  Per panel step K of the right-looking factorisation:

    owner of (K,K):      factor the diagonal block (sgpr_potrf_dev)
    process column K%pc: receives L_KK + its leaf inverses, solves its panel pieces
                         L(I,K) = A(I,K) L_KK^-T (sgpr_trsm_rlt_dev)
    process rows:        rank (q, K%pc) broadcasts its solved pieces L(I,K), I = q mod pr, along
                         process row q  -> every rank has the ROW operand of its update
    process columns:     rank (q, j) hands the blocks J = q mod pr, J = j mod pc of that piece down
                         process column j -> every rank has the COLUMN operand L(J,K), J = j mod pc
                         (RCCL broadcasts on row / column sub-communicators over xGMI: a rank
                         receives n/pr + n/pc rows of the panel, not all n)
    everyone:            local trailing update A(I,J) -= L(I,K) L(J,K)^T on the blocks it owns
                         (sgpr_gemm_nt_bc_dev, fp64 MFMA, tiles above the global diagonal
                         skipped) -- block column K+1 first, so that panel K+1 can be factored
                         and broadcast (asynchronously) underneath the bulk of update K

The Gram build needs no communication: every rank evaluates exactly the pairs of the blocks it
owns (inputs are replicated, 16 N bytes).  The triangular solves keep b replicated and exchange
one nb-vector reduce + one broadcast per block step.

This mirrors, for N beyond one GPU's HBM, the body of nll_chol (python/functions/func.py:189-196
of the reference), which has no parallel form of its own.

Storage is PACKED LOWER by blocks (round 5): of its local column block J a rank keeps only the row blocks I >= J, as one
column-major panel of its own leading dimension -- half the bytes of the dense local piece, which is what puts the
n = 262 144 configuration on two GPUs (137 GB of matrix per rank instead of 275) -- and the Gram build evaluates only
those blocks.  The factor never reads anything else.

The numerical work is delegated to an `ops` object: `HipOps` (below) binds the C ABI on torch
CUDA tensors and is the only product backend; tests/ supply a NumPy backend to exercise the
distribution logic under gloo without a GPU.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L


def grid_shape(world):
    """pr x pc process grid, pr >= pc, as square as the factorisation allows."""
    pc = int(math.isqrt(world))
    while world % pc:
        pc -= 1
    return world // pc, pc


def _count_le(K, p, nproc):
    """number of block indices I = p, p+nproc, ... with I <= K"""
    return 0 if K < p else (K - p) // nproc + 1


def hbm_plan(N, d, nb, world, rank=0):
    """Bytes of HBM rank `rank` of `world` needs for N points, d pairs per point, block size nb: the local piece of Ky (packed
    lower: of column block J the row blocks I >= J), the two sets of panel operand buffers, the diagonal block + factor
    workspace + the store of the owned diagonal blocks' workspaces, the replicated vectors.  Pure arithmetic (the same
    bookkeeping as DistFit.__init__ / _buffers), so that a run can be refused BEFORE anything is allocated."""
    pr, pc = grid_shape(world)
    pi, pj = rank % pr, rank // pr
    n = 2 * d * N
    nbk = n // nb
    rows, cols = list(range(pi, nbk, pr)), list(range(pj, nbk, pc))
    per_q = [len([J for J in cols if J % pr == q]) for q in range(pr)]
    blocks = 2 * (max(len(rows), 1) + max(len(cols), 1) + sum(max(c, 1) for c in per_q))
    leaves = (nb + 127) // 128
    work = leaves * 128 * 128 * 8 + 4 * leaves * 296 + 2 * nb * 8 + (2 << 20)      # inverses, hand-off words, solve vectors, slack
    owned_diag = len([K for K in range(nbk) if K % pr == pi and K % pc == pj])
    packed = sum(len(rows) - _count_le(J - 1, pi, pr) for J in cols)
    out = {"matrix": 8 * packed * nb * nb, "panel_buffers": 8 * blocks * nb * nb,
           "diag_block_and_workspace": 8 * nb * nb + (2 + owned_diag) * work, "vectors": 8 * 6 * n}
    out["total"] = sum(out.values())
    return out


class HipOps:
    """Block operations on torch CUDA tensors through libsympgpr_hip.so (device pointers)."""

    device_type = "cuda"

    def __init__(self, device):
        self.lib = L.load_library()
        self.device = device
        L.check(self.lib.sgpr_set_device(device.index or 0))

    def empty(self, n, dtype=torch.float64):
        return torch.empty(n, dtype=dtype, device=self.device)

    def zeros(self, n, dtype=torch.float64):
        return torch.zeros(n, dtype=dtype, device=self.device)

    @staticmethod
    def _p(t, off=0):
        return C.c_void_p(t.data_ptr() + 8 * off)

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def work_size(self, nb):
        return (self.lib.sgpr_potrf_workspace(nb) + 7) // 8  # in doubles

    def inv_size(self, nb):
        """leading doubles of the factor workspace that hold the leaf inverses: all a peer needs beside L_KK"""
        return self.lib.sgpr_potrf_inverses_bytes(nb) // 8

    def solve_status(self, nb, Lkk, work):
        """0, or 1 when a one-launch triangular solve on this workspace gave up on a hand-off (waits for the stream)"""
        rc = self.lib.sgpr_solve_status_dev(nb, self._p(Lkk), nb, self._p(work), self.stream())
        if rc == L.E_HIP:
            return 1
        L.check(rc, "sgpr_solve_status_dev")
        return 0

    def gram_nd(self, fam, d, mi, mj, Xb, Xa, hyp, A, ld):
        """(2d)^2 blocks of mi x mj for d canonical pairs; Xb (mi x 2d), Xa (mj x 2d) column-major."""
        hyp = L.f64(hyp)
        L.check(self.lib.sgpr_gram_nd_dev(L.family_id(fam), d, mi, mj, self._p(Xb), mi, self._p(Xa), mj, L.dptr(hyp),
                                          len(hyp), self._p(A), ld, mi, mj, 0, 0.0, self.stream()), "sgpr_gram_nd_dev")

    def gram_nd_sel(self, fam, d, mi, mj, Xb, Xa, hyp, A, ld, roff, coff):
        """the same pairs, only the blocks (a, b) with roff[a] >= 0 and coff[b] >= 0, at A + roff[a] + coff[b] * ld"""
        hyp = L.f64(hyp)
        ro, co = (C.c_long * (2 * d))(*roff), (C.c_long * (2 * d))(*coff)
        L.check(self.lib.sgpr_gram_nd_sel_dev(L.family_id(fam), d, mi, mj, self._p(Xb), mi, self._p(Xa), mj, L.dptr(hyp),
                                              len(hyp), self._p(A), ld, ro, co, self.stream()), "sgpr_gram_nd_sel_dev")

    def gram_pairs(self, fam, mi, mj, xb, yb, xa, ya, hyp, A, offs, ld, flags):
        """offs: element offsets of the qq / Pq / qP / PP parts inside A (or None)."""
        hyp = L.f64(hyp)
        ptr = [self._p(A, o) if o is not None else None for o in offs]
        L.check(self.lib.sgpr_gram_pairs_dev(L.family_id(fam), mi, mj, self._p(xb), self._p(yb), self._p(xa),
                                             self._p(ya), L.dptr(hyp), len(hyp), ptr[0], ptr[1], ptr[2], ptr[3],
                                             ld, 0, 0.0, flags, self.stream()), "sgpr_gram_pairs_dev")

    def potrf(self, nb, A, work, info):
        L.check(self.lib.sgpr_potrf_dev(nb, self._p(A), nb, self._p(work), 8 * work.numel(), self._p(info),
                                        self.stream()), "sgpr_potrf_dev")

    def trsm(self, m, nb, Lkk, work, B, boff, ldb):
        L.check(self.lib.sgpr_trsm_rlt_dev(m, nb, self._p(Lkk), nb, self._p(B, boff), ldb, self._p(work),
                                           self.stream()), "sgpr_trsm_rlt_dev")

    def gemm_nt(self, m, n, k, alpha, A, aoff, lda, B, boff, ldb, beta, Cm, coff, ldc):
        """C (m x n) := beta C + alpha A (m x k) B (n x k)^T on the fp64 MFMA kernel"""
        L.check(self.lib.sgpr_gemm_nt_dev(m, n, k, alpha, self._p(A, aoff), lda, self._p(B, boff), ldb, beta,
                                          self._p(Cm, coff), ldc, 0, 0, self.stream()), "sgpr_gemm_nt_dev")

    def gemm_nn(self, m, n, k, alpha, A, aoff, lda, B, boff, ldb, beta, Cm, coff, ldc):
        """C (m x n) := beta C + alpha A (m x k) B (k x n)"""
        L.check(self.lib.sgpr_gemm_nn_dev(m, n, k, alpha, self._p(A, aoff), lda, self._p(B, boff), ldb, beta,
                                          self._p(Cm, coff), ldc, self.stream()), "sgpr_gemm_nn_dev")

    def trsm_rows(self, m, nb, Lkk, work, B, boff, ldb, trans):
        """right-hand sides as the m ROWS of B (m x nb): B := B L^-T (trans = 0: L X = B^T) or B L^-1 (trans = 1: L^T X = B^T)"""
        fn = self.lib.sgpr_trsm_rl_dev if trans else self.lib.sgpr_trsm_rlt_dev
        L.check(fn(m, nb, self._p(Lkk), nb, self._p(B, boff), ldb, self._p(work), self.stream()), "sgpr_trsm_rl(t)_dev")

    def copy_blocks(self, rows, cols, cnt, src, soff, lds, sstep, dst, doff, ldd, dstep):
        """cnt blocks of rows x cols doubles, block i from src + soff + i * sstep (ld lds) to dst + doff + i * dstep (ld ldd):
        the pack / regroup copies of the panel exchange as ONE launch on the current stream"""
        L.check(self.lib.sgpr_copy_blocks_dev(rows, cols, cnt, self._p(src, soff), lds, sstep, self._p(dst, doff), ldd, dstep,
                                              self.stream()), "sgpr_copy_blocks_dev")

    def trsv(self, nb, Lkk, work, b, trans):
        L.check(self.lib.sgpr_trsv_dev(nb, self._p(Lkk), nb, self._p(work), self._p(b), trans, self.stream()),
                "sgpr_trsv_dev")

    def gemv_sub(self, trans, m, k, A, aoff, lda, x, y):
        L.check(self.lib.sgpr_gemv_sub_dev(trans, m, k, self._p(A, aoff), lda, self._p(x), self._p(y),
                                           self.stream()), "sgpr_gemv_sub_dev")

    def sync(self):
        torch.cuda.synchronize(self.device)

    def mem_free(self):
        return torch.cuda.mem_get_info(self.device)[0]

    def side(self):
        """Context of a second HIP stream: the column exchange of panel K+1 is packed and issued there,
        behind the row broadcast it depends on, while this stream runs the bulk of update K."""
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(self._side)

    def join_side(self):
        if hasattr(self, "_side"):
            torch.cuda.current_stream(self.device).wait_stream(self._side)


class DistFit:
    """Ky = build_K(x,x) + |sig2n| I, L, alpha, nll on a pr x pc grid.  All ranks call every
    method collectively."""

    _BIG = 1 << 60

    def __init__(self, ops, family, x, y, z, hyp, sig2n, nb=1024, group=None, X=None, serial=None):
        """x, y: the (q, P) coordinates of the reference's one-pair layout; or X (N x 2d) for d
        canonical pairs per point (then x, y are ignored and hyp = (lq.., lP.., sig)).
        serial (default: SGPR_DIST_SERIAL=1 in the environment): every collective is issued BLOCKING on the compute stream --
        no side stream, no asynchronous handles, one collective in flight at a time -- for telling an RCCL-only failure of
        the overlapped form from a logic error in one run."""
        import os
        self.serial = (os.environ.get("SGPR_DIST_SERIAL", "0") not in ("", "0")) if serial is None else bool(serial)
        self.ops, self.family = ops, family
        self.X = None if X is None else np.asfortranarray(X, dtype=np.float64)
        self.d = 1 if X is None else self.X.shape[1] // 2
        if X is not None:
            x = self.X[:, 0]
            y = self.X[:, self.d]
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.group = group
        self.pr, self.pc = grid_shape(self.world)
        self.pi, self.pj = self.rank % self.pr, self.rank // self.pr
        self.N = len(x)
        self.n = 2 * self.d * self.N
        # Block size: the largest divisor of N that does not exceed the request and is a multiple of the
        # 128-row leaf (any divisor for small test problems); any N / nb, for one pair per point and for d > 1
        # alike (see build()).
        self.nb = self._pick_nb(self.N, nb, 1)
        nb = self.nb
        self.nbk = self.n // nb
        self.hyp = np.asarray(hyp, dtype=np.float64)
        self.sig2n = abs(float(sig2n))
        self.rows = list(range(self.pi, self.nbk, self.pr))   # global block rows held here
        self.cols = list(range(self.pj, self.nbk, self.pc))
        self.mloc = len(self.rows) * nb
        # packed lower storage: of local column block lj (global J) the local row blocks li >= lifirst[lj] (global I >= J),
        # one column-major panel of leading dimension ld[lj] at element offset coloff[lj]
        self.lifirst = [_count_le(J - 1, self.pi, self.pr) for J in self.cols]
        self.ld = [(len(self.rows) - f) * nb for f in self.lifirst]
        self.coloff = [0]
        for v in self.ld:
            self.coloff.append(self.coloff[-1] + v * nb)
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        # HBM plan of this rank against what the device has free, BEFORE the first allocation (a rank that dies in hipMalloc
        # half way through leaves the others in a collective) -- and the verdict is SHARED before anybody raises: free memory
        # differs from device to device, and one rank raising alone would leave the others in the first broadcast of factor()
        self.plan = hbm_plan(self.N, self.d, nb, self.world, self.rank)
        assert self.plan["matrix"] == 8 * self.coloff[-1]
        if hasattr(ops, "mem_free"):
            free = ops.mem_free()
            short = torch.tensor([max(0.0, float(self.plan["total"] - free))], dtype=torch.float64, device=ops.device)
            if self.world > 1:
                dist.all_reduce(short, op=dist.ReduceOp.MAX, group=group)
            if float(short.item()) > 0:
                raise MemoryError("rank %d of %d: n = %d in %d x %d blocks on a %d x %d grid needs %.1f GB of HBM here (%s), %.1f GB are "
                                  "free; the rank that is shortest is short of %.1f GB (every rank raises)"
                                  % (self.rank, self.world, self.n, nb, nb, self.pr, self.pc, self.plan["total"] / 1e9,
                                     ", ".join("%s %.1f" % (k, v / 1e9) for k, v in self.plan.items() if k != "total"), free / 1e9,
                                     float(short.item()) / 1e9))
        self.z = torch.as_tensor(np.asarray(z, dtype=np.float64)).to(ops.device)
        self.A = ops.empty(max(self.coloff[-1], 1))             # the packed local matrix, flat
        # L_KK and the factor workspace back to back: [L_KK | leaf inverses | scratch] -- the diagonal block and the
        # inverses of its leaves travel down the process column as ONE message (the first nb^2 + inv doubles)
        self.wsize = ops.work_size(nb)
        self.inv_size = ops.inv_size(nb) if hasattr(ops, "inv_size") else self.wsize
        self._kkbuf = ops.empty(nb * nb + self.wsize)
        self.Lkk = self._kkbuf[:nb * nb]
        self.wbuf = self._kkbuf[nb * nb:]
        # the workspaces of the diagonal blocks this rank owns (leaf inverses + the solves' hand-off words), one persistent store
        self.owned = [K for K in range(self.nbk) if K % self.pr == self.pi and K % self.pc == self.pj]
        self._wstore = ops.empty(max(len(self.owned), 1) * self.wsize)
        self.work = {K: self._wstore[s * self.wsize:(s + 1) * self.wsize] for s, K in enumerate(self.owned)}
        self.info_t = ops.zeros(2, dtype=torch.int32)
        # process-column / process-row groups (every rank creates all of them, same order)
        self.col_groups = [dist.new_group([q + c * self.pr for q in range(self.pr)]) for c in range(self.pc)]
        self.row_groups = [dist.new_group([r + c * self.pr for c in range(self.pc)]) for r in range(self.pr)]
        self.alpha = None
        self.nll = None
        self.info = 0

    @staticmethod
    def _pick_nb(N, want, mult):
        """largest nb <= want with N % nb == 0 and (N / nb) % mult == 0, preferring multiples of 128"""
        cands = [b for b in range(1, min(want, N) + 1) if N % b == 0 and (N // b) % mult == 0]
        if not cands:
            raise ValueError("no block size <= %d divides N = %d with N/nb a multiple of %d" % (want, N, mult))
        aligned = [b for b in cands if b % 128 == 0]
        return max(aligned) if aligned else max(cands)

    def grank(self, pi, pj):
        return pi + pj * self.pr

    # ---- the packed layout
    def _off(self, li, lj):
        """element offset of local block (li, lj); li >= lifirst[lj]"""
        return self.coloff[lj] + (li - self.lifirst[lj]) * self.nb

    def _blk(self, li, lj):
        """local block (li, lj) as a [column, row] view"""
        nb = self.nb
        return self.A.as_strided((nb, nb), (self.ld[lj], 1), self._off(li, lj))

    def describe(self, steps=2):
        """What this rank will do, as data: its communicators and, for the first `steps` panel steps, the collectives it
        issues in issue order -- (step K, communicator, root (global rank), doubles, blocking?, stream).  Every member of a
        communicator derives the same sub-sequence for it (issue order); bench.py prints it per rank at start-up."""
        pr, pc, pi, pj, nb = self.pr, self.pc, self.pi, self.pj, self.nb
        out = {"rank": self.rank, "world": self.world, "grid": [pr, pc], "coords": [pi, pj], "block": nb, "blocks": self.nbk,
               "serial": self.serial, "storage": "packed lower by blocks",
               "row_communicator": [self.grank(pi, c) for c in range(pc)], "col_communicator": [self.grank(q, pj) for q in range(pr)],
               "hbm_plan_gb": {k: round(v / 1e9, 3) for k, v in self.plan.items()}, "steps": []}
        for K in range(min(steps, self.nbk)):
            kI, kJ = K % pr, K % pc
            seq = []
            if pj == kJ and pr > 1:
                seq.append(("col", self.grank(kI, kJ), nb * nb + self.inv_size, True, "compute"))
            nrow_blk = len(self.rows) - _count_le(K, pi, pr)
            if nrow_blk > 0 and pc > 1:
                seq.append(("row", self.grank(pi, kJ), nrow_blk * nb * nb, self.serial, "compute"))
            if pr > 1:
                for (q, _t, _p, cnt, _pos0, _ps) in self._col_plan(K)[2]:
                    seq.append(("col", self.grank(q, pj), cnt * nb * nb, self.serial, "compute" if (self.serial or K == 0) else "side"))
            out["steps"].append({"K": K, "collectives": seq})
        return out

    # ------------------------------------------------------------------ Gram build (no comm)
    def build(self):
        """Each rank evaluates the pairs of its own blocks ON OR BELOW the global block diagonal (inputs replicated, no
        collective), column block by column block into the packed panels.  Global block row I < N/nb (per coordinate for
        d > 1) holds the rows of one coordinate of the points of point-block I mod (N/nb); a column block belongs to one
        coordinate and one point-block, the rows under it to several coordinates with their own point selections: one launch
        per (column block, row coordinate) writes that part (sgpr_gram_pairs_dev takes the four parts of the one-pair layout
        one by one, sgpr_gram_nd_sel_dev the blocks (a, b) of a selection)."""
        ops, nb, N = self.ops, self.nb, self.N
        nbN = N // nb
        D = 2 * self.d
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(ops.device)
        sel = lambda blocks: np.concatenate([np.arange(B * nb, (B + 1) * nb) for B in blocks])
        cache = {}

        def pts(blocks):
            """device copies of the coordinates of a selection of point-blocks (cached: the selections repeat from column to column)"""
            key = tuple(blocks)
            if key not in cache:
                idx = sel(blocks)
                if self.X is not None:
                    cache[key] = (dev(np.asfortranarray(self.X[idx]).T.copy()).reshape(-1), None)   # (m x 2d) column-major, flat
                else:
                    cache[key] = (dev(self.x[idx]), dev(self.y[idx]))
                if len(cache) > 4 * D + 8:
                    cache.pop(next(iter(cache)))
            return cache[key]

        part = {(0, 0): (0, L.G_QQ), (1, 0): (1, L.G_PQ), (0, 1): (2, L.G_QP), (1, 1): (3, L.G_PP)}
        for lj, J in enumerate(self.cols):
            lf = self.lifirst[lj]
            if lf >= len(self.rows):
                continue
            b, Jb = J // nbN, J % nbN                          # the column's coordinate and point-block
            ca = pts([Jb])
            pos = 0                                            # row blocks of the panel written so far
            for a in range(D):
                mine = [I - a * nbN for I in self.rows[lf:] if I // nbN == a]
                if not mine:
                    continue
                rb = pts(mine)
                mi = len(mine) * nb
                off = self.coloff[lj] + pos * nb
                if self.X is not None:
                    roff = [0 if aa == a else -1 for aa in range(D)]
                    coff = [0 if bb == b else -1 for bb in range(D)]
                    ops.gram_nd_sel(self.family, self.d, mi, nb, rb[0], ca[0], self.hyp, self.A[off:], self.ld[lj], roff, coff)
                else:
                    k, flag = part[(a, b)]
                    po = [None] * 4
                    po[k] = off
                    ops.gram_pairs(self.family, mi, nb, rb[0], rb[1], ca[0], ca[1], self.hyp, self.A, po, self.ld[lj], flag)
                pos += len(mine)
        # |sig2n| on the global diagonal: the diagonal blocks this rank owns
        for K in self.owned:
            self._blk(K // self.pr, K // self.pc).diagonal().add_(self.sig2n)

    # ------------------------------------------------------------------ factorisation
    def _buffers(self, K):
        """Operand buffers of panel K (two sets: panel K+1 is in flight while update K runs)."""
        if not hasattr(self, "_bufs"):
            mk = lambda cnt: self.ops.empty(max(cnt, 1) * self.nb * self.nb)
            per_q = [len([J for J in self.cols if J % self.pr == q]) for q in range(self.pr)]
            self._bufs = [{"row": mk(len(self.rows)), "col": mk(len(self.cols)), "stage": [mk(c) for c in per_q]}
                          for _ in range(2)]
        return self._bufs[K & 1]

    def _panel_start(self, K):
        """Factor the diagonal block K, solve its panel, START the broadcast of this process row's
        solved pieces along the process row (asynchronous).  Returns (row piece, handles)."""
        ops, nb, pr, pc, pi, pj = self.ops, self.nb, self.pr, self.pc, self.pi, self.pj
        kI, kJ = K % pr, K % pc
        lj_K = K // pc
        if (pi, pj) == (kI, kJ):
            blk = self._blk(K // pr, lj_K)
            self.Lkk.view(nb, nb).copy_(blk)
            ops.potrf(nb, self.Lkk, self.wbuf, self.info_t)
            # LAPACK-style global index of the first failing minor, tracked on the device; a negative
            # value (an internal error of the factor kernel) is kept as it is
            i64 = self.info_t[:1].to(torch.int64)
            cand = torch.where(i64 > 0, i64 + K * nb, torch.where(i64 < 0, i64, torch.full_like(i64, self._BIG)))
            self.fail_t = torch.minimum(self.fail_t, cand)
            blk.copy_(self.Lkk.view(nb, nb))
            self.work[K].copy_(self.wbuf)             # (the whole workspace: the solves keep their hand-off words in it)
        li0 = _count_le(K, pi, pr)            # first local block row with I > K
        m_p = self.mloc - li0 * nb
        if pj == kJ:
            src = self.grank(kI, kJ)
            if pr > 1:
                msg = self._kkbuf[:nb * nb + self.inv_size]        # L_KK + its leaf inverses: one message
                dist.broadcast(msg, src=src, group=self.col_groups[kJ])
                self.comm_bytes += 8 * msg.numel() * (self.rank != src)
            if m_p > 0:
                ops.trsm(m_p, nb, self.Lkk, self.wbuf, self.A, self._off(li0, lj_K), self.ld[lj_K])
        nrow_blk = len(self.rows) - li0
        Lrow = self._buffers(K)["row"][:nrow_blk * nb * nb]
        handles = []
        if nrow_blk > 0:
            if pj == kJ:                      # the solved rows of the panel, packed to leading dimension m_p (one launch)
                ops.copy_blocks(m_p, nb, 1, self.A, self._off(li0, lj_K), self.ld[lj_K], 0, Lrow, 0, m_p, 0)
            if pc > 1:
                h = dist.broadcast(Lrow, src=self.grank(pi, kJ), group=self.row_groups[pi], async_op=not self.serial)
                if not self.serial:
                    handles.append(h)
                self.comm_bytes += 8 * Lrow.numel() * (pj != kJ)
        return (Lrow, nrow_blk, li0), handles

    def _col_plan(self, K):
        """For each process row q: which of my column blocks J > K it supplies (J = q mod pr), as
        (q, first position t_q in my column list, stride, count, first position in q's row piece, stride)."""
        pr, pc, pj = self.pr, self.pc, self.pj
        lj0 = _count_le(K, pj, pc)            # first local block column with J > K
        ncol_blk = len(self.cols) - lj0
        plan = []
        if ncol_blk > 0:
            J0 = self.cols[lj0]
            g = math.gcd(pc, pr)
            period, pstep = pr // g, pc // g
            for q in range(pr):
                if (q - J0) % g:
                    continue
                t_q = next(t for t in range(period) if (J0 + pc * t) % pr == q)
                cnt = len(range(t_q, ncol_blk, period))
                if cnt:
                    pos0 = (J0 + pc * t_q) // pr - _count_le(K, q, pr)
                    plan.append((q, t_q, period, cnt, pos0, pstep))
        return lj0, ncol_blk, plan

    def _col_exchange_start(self, K, rowpiece):
        """Second half of the panel exchange: every rank of process row q now holds L(I,K), I = q mod pr;
        rank (q, pj) hands the blocks that are also column blocks of process column pj down that column.
        Called behind the row broadcast (which it reads); returns (state, handles)."""
        nb, pr, pi = self.nb, self.pr, self.pi
        Lrow, nrow_blk, _ = rowpiece
        lj0, ncol_blk, plan = self._col_plan(K)
        buf = self._buffers(K)
        handles, stages = [], []
        for (q, t_q, period, cnt, pos0, pstep) in plan:
            st = buf["stage"][q][:cnt * nb * nb]
            if pi == q:                       # every pstep-th block of my row piece, packed (one launch)
                self.ops.copy_blocks(nb, nb, cnt, Lrow, pos0 * nb, nrow_blk * nb, pstep * nb, st, 0, cnt * nb, nb)
            if pr > 1:
                h = dist.broadcast(st, src=self.grank(q, self.pj), group=self.col_groups[self.pj], async_op=not self.serial)
                if not self.serial:
                    handles.append(h)
                self.comm_bytes += 8 * st.numel() * (pi != q)
            stages.append((st, t_q, period, cnt))
        return (lj0, ncol_blk, stages, buf), handles

    def _col_finish(self, state):
        """Regroup the received blocks by local column: one launch per source process row (no index tensors, no host sync)."""
        nb = self.nb
        lj0, ncol_blk, stages, buf = state
        if ncol_blk == 0:
            return None, 0, lj0
        Lcol = buf["col"][:ncol_blk * nb * nb]
        for (st, t_q, period, cnt) in stages:
            self.ops.copy_blocks(nb, nb, cnt, st, 0, cnt * nb, nb, Lcol, t_q * nb, ncol_blk * nb, period * nb)
        return Lcol, ncol_blk, lj0

    def _exchange(self, K):
        """panel K: factor / solve / both broadcast phases, waited for (used for panel 0)."""
        rowp, h = self._panel_start(K)
        for x in h:
            x.wait()
        state, h2 = self._col_exchange_start(K, rowp)
        for x in h2:
            x.wait()
        Lcol, ncol_blk, lj0 = self._col_finish(state)
        Lrow, nrow_blk, _ = rowp
        if nrow_blk == 0 or ncol_blk == 0:
            return None
        return Lrow, nrow_blk, Lcol, ncol_blk, lj0

    def _update(self, K, opnd, c_from, c_to):
        """A(I,J) -= L(I,K) L(J,K)^T on the local blocks with I >= J > K of the local column blocks
        lj0 + c_from .. lj0 + c_to - 1: one product per column block (its panel has its own leading dimension)."""
        if opnd is None:
            return
        Lrow, nrow_blk, Lcol, ncol_blk, lj0 = opnd
        c_to = min(c_to, ncol_blk)
        nb = self.nb
        li0 = _count_le(K, self.pi, self.pr)
        for c in range(c_from, c_to):
            lj = lj0 + c
            lf = self.lifirst[lj]                 # (>= li0: J > K)
            m = (len(self.rows) - lf) * nb
            if m > 0:
                self.ops.gemm_nt(m, nb, nb, -1.0, Lrow, (lf - li0) * nb, nrow_blk * nb, Lcol, c * nb, ncol_blk * nb, 1.0,
                                 self.A, self.coloff[lj], self.ld[lj])

    def factor(self):
        """Right-looking with one step of look-ahead: while the bulk of trailing update K runs,
        panel K+1 (already updated) is factored, solved and on its way to the other ranks: its row
        broadcast starts at once, its column exchange behind it on a side stream."""
        self.info = 0
        self.comm_bytes = 0
        self.fail_t = torch.full((1,), self._BIG, dtype=torch.int64, device=self.info_t.device)
        side = None if self.serial else getattr(self.ops, "side", None)      # serial: everything on the compute stream
        opnd = self._exchange(0)
        for K in range(self.nbk):
            if K + 1 < self.nbk:
                owns_next = (K + 1) % self.pc == self.pj   # block column K+1 is my first column > K
                if owns_next:
                    self._update(K, opnd, 0, 1)
                rowp, h1 = self._panel_start(K + 1)
                if side is not None:
                    with side():                           # behind the row broadcast, beside the bulk update
                        for x in h1:
                            x.wait()
                        state, h2 = self._col_exchange_start(K + 1, rowp)
                    self._update(K, opnd, 1 if owns_next else 0, 1 << 30)
                    self.ops.join_side()
                else:
                    for x in h1:
                        x.wait()
                    state, h2 = self._col_exchange_start(K + 1, rowp)
                    self._update(K, opnd, 1 if owns_next else 0, 1 << 30)
                for x in h1 + h2:
                    x.wait()
                Lcol, ncol_blk, lj0 = self._col_finish(state)
                Lrow, nrow_blk, _ = rowp
                opnd = None if (nrow_blk == 0 or ncol_blk == 0) else (Lrow, nrow_blk, Lcol, ncol_blk, lj0)
            else:
                self._update(K, opnd, 0, 1 << 30)
        t = self.fail_t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        v = int(t.item())
        if v < 0:
            raise L.SympGPRError("the factor kernel reported an internal error (%d) on some rank" % v)
        self.info = 0 if v >= self._BIG else v
        return self.info

    # ------------------------------------------------------------------ solves
    def _diag_block(self, K):
        nb = self.nb
        self.Lkk.view(nb, nb).copy_(self._blk(K // self.pr, K // self.pc))
        return self.Lkk

    def solve(self):
        """alpha = L^-T L^-1 z (replicated result), nll = z.alpha/2 + sum log diag L."""
        ops, nb, pr, pc, pi, pj = self.ops, self.nb, self.pr, self.pc, self.pi, self.pj
        b = self.z.clone()
        b2 = b.view(self.nbk, nb)
        pend = ops.zeros(self.mloc)
        vec = ops.empty(nb)
        logdet = ops.zeros(1)
        gave_up = 0          # strip solves on this rank that ran out of patience on a hand-off (see _solve_status)
        # forward: y_K = L_KK^-1 (b_K - sum_{J<K} L(K,J) y_J)
        for K in range(self.nbk):
            kI, kJ = K % pr, K % pc
            owner = self.grank(kI, kJ)
            if pi == kI:
                li_K = K // pr
                vec.copy_(pend[li_K * nb:(li_K + 1) * nb])
                dist.reduce(vec, dst=owner, op=dist.ReduceOp.SUM, group=self.row_groups[kI])
            if self.rank == owner:
                vec.neg_().add_(b2[K])
                Lkk = self._diag_block(K)
                logdet += torch.log(Lkk.view(nb, nb).diagonal()).sum()
                ops.trsv(nb, Lkk, self.work[K], vec, 0)
                gave_up += self._solve_status(nb, Lkk, self.work[K])
            dist.broadcast(vec, src=owner, group=self.group)
            b2[K].copy_(vec)
            if pj == kJ:
                li0 = _count_le(K, pi, pr)
                m_p = self.mloc - li0 * nb
                if m_p > 0:
                    # pend[rows I > K] += L(I,K) y_K   (gemv_sub subtracts: feed -y)
                    lj = K // pc
                    ops.gemv_sub(0, m_p, nb, self.A, self._off(li0, lj), self.ld[lj], -vec, pend[li0 * nb:])
        # backward: x_K = L_KK^-T (y_K - sum_{I>K} L(I,K)^T x_I)
        rows_t = torch.as_tensor(self.rows, device=b.device)
        for K in range(self.nbk - 1, -1, -1):
            kI, kJ = K % pr, K % pc
            owner = self.grank(kI, kJ)
            if pj == kJ:
                vec.zero_()
                li0 = _count_le(K, pi, pr)
                m_p = self.mloc - li0 * nb
                if m_p > 0:
                    xloc = b2[rows_t[li0:]].reshape(-1).contiguous()
                    # vec -= L_panel^T xloc  -> vec = -(sum)
                    lj = K // pc
                    ops.gemv_sub(1, m_p, nb, self.A, self._off(li0, lj), self.ld[lj], xloc, vec)
                dist.reduce(vec, dst=owner, op=dist.ReduceOp.SUM, group=self.col_groups[kJ])
            if self.rank == owner:
                vec.add_(b2[K])
                ops.trsv(nb, self._diag_block(K), self.work[K], vec, 1)
                gave_up += self._solve_status(nb, self.Lkk, self.work[K])
            dist.broadcast(vec, src=owner, group=self.group)
            b2[K].copy_(vec)
        # one reduction carries the log-determinant and the count of solves that gave up (never silently NaN)
        tail = torch.cat([logdet, torch.full_like(logdet, float(gave_up))])
        dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
        logdet = tail[:1]
        if float(tail[1].item()) > 0:
            raise L.SympGPRError("a triangular solve gave up on a hand-off between two workgroups on some rank")
        self.alpha = b
        self.nll = float((0.5 * torch.dot(self.z, b) + logdet[0]).item())
        return b

    def solve_rhs(self, B):
        """X = L^-T L^-1 B for a block of right-hand sides B (n x nrhs; NumPy array or tensor, the same on every rank) against
        the distributed factor: what the reference does with matmul(Kyinv, .) at every prediction
        (python/05_tokamak/SympGPR/sympgpr.f90:72,85,121) when Kyinv cannot exist on one device.  Returns X (n x nrhs,
        replicated) as a tensor on the ops' device.

        The right-hand sides are kept as ROWS (B^T, nrhs x n column-major): block K of all of them is one contiguous piece, so a
        block step exchanges the WHOLE block with one reduce to the owner of L_KK and one broadcast, as solve() does for one
        vector; the local work is two matrix products per block step on the fp64 MFMA kernel (sgpr_gemm_nt_dev /
        sgpr_gemm_nn_dev) and the owner's triangular solve with L_KK (sgpr_trsm_rlt_dev / sgpr_trsm_rl_dev)."""
        ops, nb, pr, pc, pi, pj = self.ops, self.nb, self.pr, self.pc, self.pi, self.pj
        Bt = torch.as_tensor(np.asarray(B, dtype=np.float64) if not torch.is_tensor(B) else B).to(ops.device)
        if Bt.dim() == 1:
            Bt = Bt.reshape(-1, 1)
        if Bt.shape[0] != self.n:
            raise ValueError("solve_rhs: B has %d rows, the factor is of order %d" % (Bt.shape[0], self.n))
        m = int(Bt.shape[1])
        # [n, m] row-major = (m x n) column-major with leading dimension m: the right-hand sides as rows
        X = Bt.contiguous().clone().reshape(-1)
        XK = lambda K: X[K * nb * m:(K + 1) * nb * m]            # block K of every right-hand side: (m x nb), contiguous
        pend = ops.zeros(m * self.mloc)                          # (m x mloc): what the solved blocks contribute to my rows
        vec = ops.empty(m * nb)
        # forward: Y_K = (B_K - sum_{J<K} Y_J L(K,J)^T) L_KK^-T
        for K in range(self.nbk):
            kI, kJ = K % pr, K % pc
            owner = self.grank(kI, kJ)
            if pi == kI:
                li_K = K // pr
                vec.copy_(pend[li_K * nb * m:(li_K + 1) * nb * m])
                dist.reduce(vec, dst=owner, op=dist.ReduceOp.SUM, group=self.row_groups[kI])
            if self.rank == owner:
                vec.neg_().add_(XK(K))
                ops.trsm_rows(m, nb, self._diag_block(K), self.work[K], vec, 0, m, 0)
            dist.broadcast(vec, src=owner, group=self.group)
            XK(K).copy_(vec)
            if pj == kJ:
                li0 = _count_le(K, pi, pr)
                m_p = self.mloc - li0 * nb
                if m_p > 0:
                    # pend[:, rows I > K] += Y_K L(I,K)^T
                    lj = K // pc
                    ops.gemm_nt(m, m_p, nb, 1.0, vec, 0, m, self.A, self._off(li0, lj), self.ld[lj], 1.0, pend, li0 * nb * m, m)
        # backward: X_K = (Y_K - sum_{I>K} X_I L(I,K)) L_KK^-1
        rows_t = torch.as_tensor(self.rows, device=X.device)
        for K in range(self.nbk - 1, -1, -1):
            kI, kJ = K % pr, K % pc
            owner = self.grank(kI, kJ)
            if pj == kJ:
                vec.zero_()
                li0 = _count_le(K, pi, pr)
                m_p = self.mloc - li0 * nb
                if m_p > 0:
                    xloc = X.view(self.nbk, nb * m)[rows_t[li0:]].reshape(-1).contiguous()   # my rows of the solution so far, side by side
                    lj = K // pc
                    ops.gemm_nn(m, nb, m_p, -1.0, xloc, 0, m, self.A, self._off(li0, lj), self.ld[lj], 1.0, vec, 0, m)
                dist.reduce(vec, dst=owner, op=dist.ReduceOp.SUM, group=self.col_groups[kJ])
            if self.rank == owner:
                vec.add_(XK(K))
                ops.trsm_rows(m, nb, self._diag_block(K), self.work[K], vec, 0, m, 1)
            dist.broadcast(vec, src=owner, group=self.group)
            XK(K).copy_(vec)
        return X.view(self.n, m)

    def _solve_status(self, nb, Lkk, work):
        """the strip solves bound their waits; a solve that gave up is reported through the solve's last reduction"""
        fn = getattr(self.ops, "solve_status", None)
        return fn(nb, Lkk, work) if fn else 0

    def run(self):
        self.build()
        info = self.factor()
        if info:
            raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % info)
        return self.solve()

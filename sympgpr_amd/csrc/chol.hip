// chol.hip -- lower Cholesky factor / solves on one MI355X.
//
// Replaces scipy.linalg.cholesky(Ky, lower=True) -> LAPACK dpotrf and the two
// solve_triangular -> dtrtrs calls of python/functions/func.py:165-177,184-186,193-195.
//
// Structure (right-looking, recursively blocked): at every level
//     factor the leading block  ->  panel solve A21 := A21 L11^-T  ->  trailing update
//     A22 -= A21 A21^T (SYRK, lower tiles only)  ->  factor A22,
// with the split at a multiple of LEAF near the middle, so almost all of the n^3/3 flop are
// large-k GEMM/SYRK calls on the fp64 MFMA kernel (gemm_f64.hip) and the trailing matrix is
// re-read log2(n/LEAF) times instead of n/nb times.  The recursion bottoms out in LEAF = 128
// diagonal blocks that ONE workgroup factors entirely in LDS (128 x 129 fp64 = 129 KiB of the
// CU's 160 KiB) and then inverts in place; the inverse (LEAF x LEAF, zero upper) is kept in
// the workspace so that every panel solve and every triangular solve below is a multiply with
// inv(L_leaf) -- i.e. GEMM work on the matrix cores instead of a substitution.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "cholq.h"
#include "common.h"
#include "gemm_tile.h"
#include "leaf.h"

namespace sgpr {

namespace {

using namespace leaf;


__global__ __launch_bounds__(LT) void leaf_kernel(int nb, double *A, size_t lda, double *inv,
                                                  int *dinfo, int goff, int mode,
                                                  unsigned long long *stamps)
{
    __shared__ double s[LEAF_LDS];
    // a full leaf gets its loop bounds at compile time (59 us; with the run-time order of a partial leaf 64)
    if (nb == LEAF) leaf_body(s, (int)LEAF, A, lda, inv, dinfo, goff, mode, stamps);
    else            leaf_body(s, nb, A, lda, inv, dinfo, goff, mode, stamps);
}

// b(nb) := inv(L) b  or  inv(L)^T b   (inv: LEAF x LEAF lower, zero upper).
// 1024 threads: row i = t & 127, k-slice = t >> 7 (8 slices of 16): sixteen independent loads per
// thread instead of one 128-long dependent chain (25 us -> ~3 us per call).
constexpr int MV_T = 1024;
__global__ __launch_bounds__(MV_T) void leaf_matvec_kernel(int nb, const double *inv, double *b,
                                                           int trans)
{
    __shared__ double sb[LEAF];
    __shared__ double part[MV_T / LEAF][LEAF];
    const int t = threadIdx.x, i = t & (LEAF - 1), ks = t >> 7;
    if (t < LEAF) sb[t] = t < nb ? b[t] : 0.0;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int kk = 0; kk < LEAF / (MV_T / LEAF); ++kk) {
        const int k = ks * (LEAF / (MV_T / LEAF)) + kk;
        // inv is zero above the diagonal and outside nb x nb, so no triangular bounds are needed
        const double m = trans ? inv[k + i * LEAF] : inv[i + k * LEAF];
        acc = __builtin_fma(m, sb[k], acc);
    }
    part[ks][i] = acc;
    __syncthreads();
    if (t < nb) {
        double r = 0.0;
#pragma unroll
        for (int q = 0; q < MV_T / LEAF; ++q) r += part[q][t];
        b[t] = r;
    }
}

// b(n) := L^-1 b (trans = 0) or L^-T b (trans = 1) for a diagonal block of up to TRSV_BLOCK rows in
// ONE single-workgroup launch: the leaf-by-leaf substitution (x_j = inv(L_jj) b_j, off-diagonal
// leaf rows folded in before it) that the recursion would spread over 2 n/128 - 1 launches.  A
// triangular solve with one right-hand side is a chain of dependent ~10 us launches otherwise --
// 510 of them for n = 16384, where the solve stage was a tenth of the whole fit.
constexpr int TRSV_BLOCK = 512;
__global__ __launch_bounds__(MV_T) void trsv_block_kernel(int n, const double *L, size_t ldl,
                                                          const double *inv, double *b, int trans)
{
    __shared__ double sx[TRSV_BLOCK];               // b on entry, x as it is produced
    __shared__ double part[MV_T / LEAF][LEAF];
    const int t = threadIdx.x, i = t & (LEAF - 1), ks = t >> 7;    // 128 rows x 8 slices
    const int lane = t & 63, wave = t >> 6;
    for (int r = t; r < TRSV_BLOCK; r += MV_T) sx[r] = r < n ? b[r] : 0.0;
    __syncthreads();
    const int nl = (n + LEAF - 1) / LEAF;
    for (int jj = 0; jj < nl; ++jj) {
        const int j = trans ? nl - 1 - jj : jj;
        const int r0 = j * LEAF, nj = min(LEAF, n - r0);
        // ---- fold the already known x into b_j
        if (!trans) {
            // b_j[i] -= sum_{c < r0} L[r0 + i, c] x[c]: rows contiguous -> thread (i, slice) strides the columns
            // (r0 is a multiple of 128: 16 columns per slice and leaf, four independent chains)
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            if (i < nj) {
                const double *row = L + (size_t)(r0 + i);
                for (int c = ks; c < r0; c += 4 * (MV_T / LEAF)) {
                    const double l0 = row[(size_t)c * ldl], l1 = row[(size_t)(c + 8) * ldl];
                    const double l2 = row[(size_t)(c + 16) * ldl], l3 = row[(size_t)(c + 24) * ldl];
                    a0 = __builtin_fma(l0, sx[c], a0);
                    a1 = __builtin_fma(l1, sx[c + 8], a1);
                    a2 = __builtin_fma(l2, sx[c + 16], a2);
                    a3 = __builtin_fma(l3, sx[c + 24], a3);
                }
            }
            const double acc = (a0 + a1) + (a2 + a3);
            part[ks][i] = acc;
            __syncthreads();
            if (t < nj) {
                double r = 0.0;
#pragma unroll
                for (int q = 0; q < MV_T / LEAF; ++q) r += part[q][t];
                sx[r0 + t] -= r;
            }
        } else {
            // b_j[c] -= sum_{r >= r0 + LEAF} L[r, r0 + c] x[r]: one wave per column, lanes stride the rows
            const int rb = r0 + LEAF;
            constexpr int NWV = MV_T / 64, CPW = LEAF / NWV;   // 16 waves, 8 columns each
            double acc[CPW];
#pragma unroll
            for (int q = 0; q < CPW; ++q) acc[q] = 0.0;
            const double *col0 = L + (size_t)(r0 + wave) * ldl;
            for (int r = rb + lane; r < n; r += 64) {          // the 8 columns' loads go out together
                const double xr = sx[r];
#pragma unroll
                for (int q = 0; q < CPW; ++q) {
                    const int c = wave + q * NWV;
                    const double v = c < nj ? col0[(size_t)q * NWV * ldl + r] : 0.0;
                    acc[q] = __builtin_fma(v, xr, acc[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < CPW; ++q) {
                double a = acc[q];
                for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
                if (lane == 0) part[0][wave + q * NWV] = a;
            }
            __syncthreads();
            if (t < nj) sx[r0 + t] -= part[0][t];
        }
        __syncthreads();
        // ---- x_j = inv(L_jj) b_j  or  inv(L_jj)^T b_j  (inv is zero above the diagonal and outside nj x nj)
        const double *iv = inv + (size_t)j * LEAF * LEAF;
        double acc = 0.0;
#pragma unroll
        for (int kk = 0; kk < LEAF / (MV_T / LEAF); ++kk) {
            const int k = ks * (LEAF / (MV_T / LEAF)) + kk;
            const double m = trans ? iv[k + i * LEAF] : iv[i + k * LEAF];
            acc = __builtin_fma(m, sx[r0 + k], acc);
        }
        __syncthreads();                            // every thread has read b_j before it is overwritten
        part[ks][i] = acc;
        __syncthreads();
        if (t < nj) {
            double r = 0.0;
#pragma unroll
            for (int q = 0; q < MV_T / LEAF; ++q) r += part[q][t];
            sx[r0 + t] = r;
        }
        __syncthreads();
    }
    for (int r = t; r < n; r += MV_T) b[r] = sx[r];
}


// ---- persistent panel kernel ------------------------------------------------------------------
// One launch factors a whole block column ("panel") of the blocked right-looking driver: the
// nb x nb diagonal block (W = nb / 128 leaf columns) and all rows below it (R strips of 128 rows,
// the first W of them being the diagonal block's).  In round 1 every leaf column cost three dependent
// launches (leaf, rows-below x inverse, fold into the remaining columns: 60 + 74 + 99 us) on the
// stream the whole factorisation waits for; here the same tasks run inside ONE grid and hand their
// results over through flags in global memory:
//     diagonal strip g (its own workgroup), columns c = 0 .. g-1:
//         wait E[c];  X = A(g,c) inv(L_cc)^T by a blocked solve in the MFMA accumulators (trsm_solve: needs L_cc
//         and its 16 x 16 diagonal inverses only)                                          -> F[g][c]
//         c < g-1:  A(g,g) -= X X^T out of an LDS image of X;  wait F[c'][c];  A(g,c') -= X L(c',c)^T, c' in (c, g)
//         c = g-1:  A(g,g) -= X X^T lands in the leaf's LDS block and the strip goes straight on to
//     leaf(g): factor A(g,g) in LDS, L out write-through as its columns become final,
//         L + diagonal inverses handed over                                                -> E[g]
//         full inverse (recursive doubling)                                                -> I[g]
//     the chain the next panel step waits for is  leaf -> E -> solve -> update -> leaf  inside ONE CU's
//     registers and LDS per column (~70 us; as three launches per column it was 233 us).
//     strip r below the diagonal block (shared round-robin by the other workgroups), columns c = 0 .. W-1:
//         wait I[c];  X = A(r,c) inv(L_cc)^T  (a 128^3 product on the matrix cores);  wait F[c'][c];
//         A(r,c') -= X L(c',c)^T for c' in (c, W);  on the LAST column: wait E[c] and the blocked solve.
// Two tickets: diagonal strips go to workgroups whose id is a multiple of 8 (one XCD), in their order of arrival;
// everybody else, and the leftovers of the first kind, takes the strips below in order of arrival.
// Hand-off protocol (cdna_hip_programming.md, guideline 16): plain payload stores -> every storing
// wave drains vmcnt -> workgroup barrier -> one lane: agent-scope release fence, drain, relaxed
// agent-scope flag store (panel_publish); payloads stored write-through (sc1) skip the fence
// (panel_publish_wt, the leaf's E hand-off).  The consumer polls the word relaxed (one lane, s_sleep between
// polls), then ONE agent-scope acquire, drain, workgroup barrier, plain loads.
// Forward progress does not rely on co-residency of the grid: a workgroup only waits for flags of diagonal
// strips, whose tickets the first eligible workgroups to arrive hold, in order -- every flag a diagonal strip
// waits for is set by one that started before it.  Every spin is bounded; a timeout is reported through *dinfo
// (PANEL_TIMEOUT).
constexpr int PW_MAX = 16;                      // leaf columns per panel (nb <= 2048)
constexpr int PFLAG_STRIDE = 2 + 2 * PW_MAX + PW_MAX * PW_MAX + 6;   // ints of flag state per panel (296)
constexpr int PANEL_TIMEOUT = POTRF_HANDOFF_TIMEOUT;   // *dinfo value: a hand-off was never published
constexpr int PANEL_G_MAX = 256;                // workgroups (each holds a whole CU: 133 KiB of LDS)

struct PanelArgs {
    double *P;        // panel origin: element (k0, k0) of the matrix
    size_t lda;
    int R, W, G;      // row strips (incl. the W diagonal ones), leaf columns, workgroups (diagonal strips + strips below)
    int NH;           // helper workgroups, tickets W .. W+NH-1 (0 or (W-1)(W-2)/2): one per tile (g, c'), 1 <= c' < g < W, of the
                      // diagonal block -- it applies the columns c < c' to that tile so that strip g does not have to
    int below_early;  // strips below: every column by the blocked solve on the early hand-off (not only the last one)
    int tiles;        // rows below the diagonal block TILE by tile (r, c') instead of strip by strip: the G - W workgroups behind the
                      // helpers draw tiles in column-major order; row r's progress (columns solved) in flags[PFLAG_STRIDE + r - W]
                      // -- the words of the panel's second leaf column, which no panel kernel of this schedule uses (needs W >= 2,
                      // R - W < PFLAG_STRIDE)
    double *inv;      // leaf inverses of this panel's W leaves (LEAF x LEAF each)
    int *dinfo;
    int goff;         // global index of the panel's first row / column (LAPACK info)
    unsigned long long *dbg;   // per-leaf-column time stamps (16 each) or null; written only in -DSGPR_PANEL_DBG builds
    int *flags;       // PFLAG_STRIDE ints, zero on entry: [0] ticket, [2 + c] I[c], [2 + PW_MAX + c] E[c], [2 + 2 PW_MAX + r * PW_MAX + c] F[r][c]
    // task-queue driver (cholq.h): the trailing updates of the earlier panels arrive tile by tile instead of behind a
    // kernel boundary.  Diagonal strip g starts once its tiles (row tile ver_i0 + g / 2, column tiles ver_j0 .. ver_j0 + g)
    // carry ver_need updates; `abort` is the queue's give-up word (set here too when a hand-off times out).  Null otherwise.
    // The strips below the diagonal block (there: the rows of the NEXT diagonal block) wait the same way for their W tiles
    // and leave tver[tver_r0 + r] = ver_need + 1 behind (strip r of the panel solved against this panel).
    const int *ver;
    int ver_ld, ver_i0, ver_j0, ver_need;
    int *tver;
    int tver_r0, tver_val;
    int *abort;
    unsigned long long *census;   // diagnostics: 4 words per workgroup (place, start, end, strip), or null
};

typedef __attribute__((address_space(1))) int gint;
#ifdef SGPR_PANEL_DBG
constexpr bool PANEL_DBG = true;
#else
constexpr bool PANEL_DBG = false;    // make EXTRA=-DSGPR_PANEL_DBG + SGPR_PANEL_DBG=1: the chain's time stamps per leaf column
#endif
static unsigned long long *g_panel_dbg = nullptr;   // debug builds only (one factorisation at a time there); never read otherwise

__device__ __forceinline__ void panel_publish(int *flag)
{
    // every wave has drained its stores before the barrier; then one lane releases and signals
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store((gint *)flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ... with a value (the task-queue driver's version words)
__device__ __forceinline__ void panel_publish_val(int *word, int val)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store((gint *)word, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the same for a payload that went out with write-through (sc1) stores only: no L2 write-back to wait for
__device__ __forceinline__ void panel_publish_wt(int *flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store((gint *)flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait until every word flags[idx[0..cnt)] has reached `need`.  Returns false (workgroup-uniform) on timeout, or when
// the task-queue driver has given the factorisation up (`abort` word set; null outside that driver).
__device__ __forceinline__ bool panel_wait_ge(const int *flags, const int *idx, int cnt, int need, int *dinfo, int *sh,
                                              int *abort = nullptr)
{
    if (threadIdx.x == 0) {
        int ok = 1;
        for (int q = 0; q < cnt && ok; ++q) {
            gint *f = (gint *)(flags + idx[q]);
            unsigned spins = 0;
            const unsigned long long twait0 = abort ? __builtin_amdgcn_s_memrealtime() : 0;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(16);
                ++spins;
                if (abort) {
                    // task-queue driver: a long wait backs off to one look every ~2 us (7 us cost 1 - 2 % of the whole
                    // factorisation: the chain is its critical path), and gives up when the workers have, or after 20 s of
                    // REAL time
                    if (spins > 16) __builtin_amdgcn_s_sleep(64);
                    if ((spins & 15u) == 0) {
                        if (__hip_atomic_load((gint *)abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = -1; break; }
                        if (__builtin_amdgcn_s_memrealtime() - twait0 > 20ull * 100000000ull) { ok = 0; break; }
                    }
                } else if (spins > (3u << 20)) { ok = 0; break; }    // ~ 3 s
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ok == 0) {
            atomicCAS(dinfo, 0, PANEL_TIMEOUT);
            if (abort) {
                // task-queue driver's post-mortem words (abort = qs + Q_ABORT): who gave up waiting for what
                if (atomicCAS(abort + 7, 0, 1) == 0) {
                    abort[8] = (int)blockIdx.x; abort[9] = idx[0]; abort[10] = cnt; abort[11] = need;
                    abort[12] = (int)(flags == nullptr);
                }
                __hip_atomic_store((gint *)abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        *sh = ok > 0;
    }
    __syncthreads();
    const int ok = *sh;
    __syncthreads();
    return ok != 0;
}
__device__ __forceinline__ bool panel_wait(int *flags, const int *idx, int cnt, int *dinfo, int *sh, int *abort = nullptr)
{
    return panel_wait_ge(flags, idx, cnt, 1, dinfo, sh, abort);
}

// C (128 x 128) := beta C + alpha A B^T, k = 128, one workgroup of 256 threads, operands / result in
// global memory (column-major panels: see gemm_tile.h)
__device__ __forceinline__ void panel_product(double *smem, double alpha, const double *A, size_t lda,
                                              const double *B, size_t ldb, double beta, double *C, size_t ldc)
{
    tile::GemmArgs ga{};
    ga.m = LEAF; ga.n = LEAF; ga.k = LEAF;
    ga.alpha = alpha; ga.beta = beta;
    ga.A = A; ga.lda = lda; ga.B = B; ga.ldb = ldb; ga.C = C; ga.ldc = ldc;
    ga.stamps = nullptr;
    // the previous task's stores are visible to this workgroup, and its LDS reads are finished
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tile::gemm_body_dma<LEAF, LEAF, 2>(ga, smem, 0, 0);
}

// ---- the chain of the factorisation inside the panel kernel: solve -> update -> leaf in registers and LDS --------
// X (128 x 128, in place) := X inv(L_cc)^T WITHOUT the leaf's full inverse: a right-looking blocked solve over the
// eight 16-column blocks,  X_j = S_j inv(L_jj)^T;  S_j' -= X_j L(j',j)^T for j' > j,  that needs L_cc and the
// inverses of its 16 x 16 diagonal blocks only -- what a leaf hands over ~20 us before its full inverse (E[c]).
// The rows of X are independent, so every wave takes 32 of them and runs alone: the tile lives in its MFMA
// accumulators (row = lane & 15, column = 4 r + lane >> 4), and that layout IS the operand layout of the next
// product (k = 4 r + lane >> 4), so X_j goes from one v_mfma to the next without touching LDS; only L_cc is
// staged (its 28 blocks below the diagonal, and the inverted diagonal blocks in place of L's own).  288 MFMAs per
// wave.  The tile is fetched BEFORE the wait for E[c] (trsm_prefetch): it has been final since the previous column.
// leading dimension of the LDS images the matrix cores read their operands from here: 16 rows x 4 k-columns per
// fragment read, so the column stride must put k, k+1 32 banks apart (144 * 8 B = 2 * 576 B): conflict-free.  With the
// leaf's own 130 the same reads were 2-4-way conflicts and the X X^T update took 17 us instead of 9.
constexpr int ILD = LEAF + 16;
typedef double4_t XTile[2][8];      // this wave's 32 rows: [row block][16-column block][r]

// block t = 0..27 of the strictly lower 16 x 16 blocks of a 128 x 128 tile: t = bi (bi - 1) / 2 + bj, 0 <= bj < bi <= 7
__device__ __forceinline__ constexpr int below_bi(int t)
{
    int bi = 1;
    while ((bi + 1) * bi / 2 <= t) ++bi;
    return bi;
}
__device__ __forceinline__ constexpr int below_bj(int t) { return t - below_bi(t) * (below_bi(t) - 1) / 2; }

__device__ __forceinline__ void trsm_prefetch(const double *X, size_t lda, XTile &x)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[rb][j][r] = X[(size_t)(32 * wave + 16 * rb + l15) + (size_t)(16 * j + 4 * r + l4) * lda];
}

__device__ __forceinline__ void trsm_solve(double *s, const double *Lcc, const double *invd, size_t lda, XTile &x,
                                           unsigned long long *st = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    __syncthreads();                                   // the previous task is done with the LDS buffer
    // the 28 blocks below the diagonal ones: 14 16-byte loads per thread, ALL in flight at once (one round trip to
    // wherever the leaf's write-through stores went; in batches of 8 the staging took 8.7 us, four round trips)
    double2_t lv[14];
#pragma unroll
    for (int q = 0; q < 14; ++q) {
        // 16-byte piece e = 256 q + tid of 28 blocks x 16 columns x 8 row pairs: block 2 q + (tid >> 7), a compile-
        // time pair per q (a per-lane search for it cost 3 us: the loop diverges)
        const int col = (tid >> 3) & 15, rp = tid & 7;
        const int bi = (tid & 128) ? below_bi(2 * q + 1) : below_bi(2 * q);
        const int bj = (tid & 128) ? below_bj(2 * q + 1) : below_bj(2 * q);
        lv[q] = *reinterpret_cast<const double2_t *>(Lcc + (size_t)(16 * bi + 2 * rp) + (size_t)(16 * bj + col) * lda);
    }
    double dv[LEAF * PW / LT];
#pragma unroll
    for (int it = 0; it < LEAF * PW / LT; ++it) {
        const int idx = it * LT + tid;
        const int i = idx % LEAF, c = (i / PW) * PW + idx / LEAF;
        dv[it] = invd[(size_t)i + (size_t)c * LEAF];
    }
    if (PANEL_DBG && st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) st[-3] = __builtin_amdgcn_s_memrealtime();   // slot 3
    }
#pragma unroll
    for (int q = 0; q < 14; ++q) {
        const int col = (tid >> 3) & 15, rp = tid & 7;
        const int bi = (tid & 128) ? below_bi(2 * q + 1) : below_bi(2 * q);
        const int bj = (tid & 128) ? below_bj(2 * q + 1) : below_bj(2 * q);
        *reinterpret_cast<double2_t *>(s + (16 * bj + col) * ILD + 16 * bi + 2 * rp) = -lv[q];   // -L: S_j' += X_j (-L)^T
    }
#pragma unroll
    for (int it = 0; it < LEAF * PW / LT; ++it) {      // inv(L_jj) where L_jj would be
        const int idx = it * LT + tid;
        const int i = idx % LEAF, c = (i / PW) * PW + idx / LEAF;
        s[c * ILD + i] = dv[it];
    }
    __syncthreads();
    if (PANEL_DBG && st && tid == 0) st[0] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            double4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                y = __builtin_amdgcn_mfma_f64_16x16x4f64(s[(16 * j + 4 * kk + l4) * ILD + 16 * j + l15], x[rb][j][kk], y, 0, 0, 0);
            x[rb][j] = y;
        }
#pragma unroll
        for (int jj = j + 1; jj < 8; ++jj) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double lneg = s[(16 * j + 4 * kk + l4) * ILD + 16 * jj + l15];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    x[rb][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(lneg, x[rb][j][kk], x[rb][jj], 0, 0, 0);
            }
        }
    }
    if (PANEL_DBG && st && tid == 0) st[1] = __builtin_amdgcn_s_memrealtime();
}

// write-through (sc1) stores: the flag that hands X over needs no L2 write-back (panel_publish_wt)
__device__ __forceinline__ void trsm_store(double *X, size_t lda, const XTile &x)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                store_wt(X + (size_t)(32 * wave + 16 * rb + l15) + (size_t)(16 * j + 4 * r + l4) * lda, x[rb][j][r]);
}

// The 36 16 x 16 blocks of the lower triangle of a 128 x 128 tile, nine per wave, as block PAIRS {b, b'} of the
// eight row blocks: two {b,b}, two each with b' - b = 1, 2, 3 (mod 8) and one with b' - b = 4.  In terms of eight
// SLOTS the nine pairs are the same for every wave -- only the slot -> block map depends on the wave -- so the
// registers the MFMAs name are compile-time while the LDS / memory addresses carry the wave:
//   slot i = 0..4 -> block (2 w + i) mod 8,   slot 6 -> block w,   slot 7 -> block w + 4
// A pair whose first block is the smaller one is the TRANSPOSE of a lower block (C is symmetric): it is fetched
// and stored with rows and columns exchanged.
constexpr int DP[9][2] = {{0, 0}, {1, 1}, {1, 0}, {2, 0}, {3, 0}, {2, 1}, {3, 1}, {4, 1}, {7, 6}};   // (rows' slot, columns' slot)
typedef double4_t CTile[9];
struct DiagMap {
    int blk[8];
    __device__ __forceinline__ DiagMap(int wave)
    {
#pragma unroll
        for (int i = 0; i < 5; ++i) blk[i] = (2 * wave + i) & 7;
        blk[5] = 0; blk[6] = wave; blk[7] = wave + 4;
    }
    // Pair p in a column-major tile with leading dimension ld: element r of lane (l15, l4) sits at
    // base + voff + r * rstep, where base and rstep are wave-uniform and voff is one of two per-lane offsets.
    // (accumulator: rows <- l15 of block bp, columns <- 4 r + l4 of block bq; bp < bq: the transposed block)
    __device__ __forceinline__ void where(int p, size_t ld, int l15, int l4, size_t &base, unsigned &voff, size_t &rstep, bool &diagonal) const
    {
        const int bp = blk[DP[p][0]], bq = blk[DP[p][1]];
        const bool lower = bp >= bq;
        base = lower ? (size_t)16 * bp + (size_t)16 * bq * ld : (size_t)16 * bq + (size_t)16 * bp * ld;
        voff = lower ? (unsigned)l15 + (unsigned)l4 * (unsigned)ld : (unsigned)l4 + (unsigned)l15 * (unsigned)ld;   // < 2^31: ld <= 2^24
        rstep = lower ? 4 * ld : 4;
        diagonal = bp == bq;
    }
};

// this wave's nine blocks of C
__device__ __forceinline__ void diag_prefetch(const double *C, size_t lda, CTile &c)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const DiagMap dm(wave);
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        size_t base, rstep;
        unsigned voff;
        bool dg;
        dm.where(p, lda, l15, l4, base, voff, rstep, dg);
#pragma unroll
        for (int r = 0; r < 4; ++r) c[p][r] = (C + base + r * rstep)[voff];   // uniform pointer + 32-bit lane offset
    }
}

// C (the strip's diagonal tile) -= X X^T, lower blocks only: X goes from the solve's accumulators into LDS as a
// [k][row] image, every wave multiplies its nine blocks out of it (per k-chunk: the fragments of its seven row
// blocks, fetched one chunk ahead, then nine MFMAs on registers).
__device__ __forceinline__ void diag_multiply(double *s, const XTile &x, CTile &c, unsigned long long *st = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    __syncthreads();                                   // every wave is done with L_cc in s
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[(16 * j + 4 * r + l4) * ILD + 32 * wave + 16 * rb + l15] = x[rb][j][r];
    __syncthreads();
    if (PANEL_DBG && st && tid == 0) st[0] = __builtin_amdgcn_s_memrealtime();
    const DiagMap dm(wave);
    double f[2][8];
    auto fetch = [&](int k, double (&ff)[8]) {
        const double *row = s + (4 * k + l4) * ILD + l15;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i != 5) ff[i] = row[16 * dm.blk[i]];
    };
    fetch(0, f[0]);
#pragma unroll 4
    for (int k = 0; k < LEAF / 4; ++k) {
        if (k + 1 < LEAF / 4) fetch(k + 1, f[(k + 1) & 1]);
#pragma unroll
        for (int p = 0; p < 9; ++p)
            c[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(-f[k & 1][DP[p][1]], f[k & 1][DP[p][0]], c[p], 0, 0, 0);
    }
    if (PANEL_DBG && st && tid == 0) st[2] = __builtin_amdgcn_s_memrealtime();
}

// ... and straight into the leaf's LDS block, where leaf_body expects the tile: no trip through memory between
// the solve, the update and the factorisation of the chain
__device__ __forceinline__ void diag_update_into_leaf(double *s, const XTile &x, CTile &c, unsigned long long *st = nullptr)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    diag_multiply(s, x, c, st);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the X stores of trsm_store: drained long ago)
    __syncthreads();                                   // every wave is done with the image
    // (the blocks above the diagonal keep whatever the image left there: leaf_body never reads them -- every
    // read of s in it is on or below the diagonal, or masked by a select)
    const DiagMap dm(wave);
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        size_t base, rstep;
        unsigned voff;
        bool dg;
        dm.where(p, LLD, l15, l4, base, voff, rstep, dg);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[base + voff + r * rstep] = (!dg || l15 >= 4 * r + l4) ? c[p][r] : 0.0;
    }
    __syncthreads();
}

// the same update with the tile staying in memory (an earlier column of the strip)
__device__ __forceinline__ void diag_update_global(double *s, const XTile &x, CTile &c, double *C, size_t lda)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    diag_multiply(s, x, c);
    const DiagMap dm(wave);
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        size_t base, rstep;
        unsigned voff;
        bool dg;
        dm.where(p, lda, l15, l4, base, voff, rstep, dg);
#pragma unroll
        for (int r = 0; r < 4; ++r) (C + base + r * rstep)[voff] = c[p][r];
    }
}

// One strip of one panel (g < W: diagonal strip g; otherwise a strip below the diagonal block).  False: a hand-off timed
// out or the factorisation has been given up.  `s`: LEAF * ILD doubles of LDS, `sh`: two ints.
__device__ __forceinline__ bool panel_strip(const PanelArgs &a, const int g, double *s, int *sh)
{
    const int tid = threadIdx.x;
    const int W = a.W, R = a.R, G = a.G;
    unsigned long long *const cen = (a.census && blockIdx.x < 32) ? a.census + 4 * blockIdx.x : nullptr;
    if (cen && tid == 0) {
        cen[0] = ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) |
                 (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        cen[1] = __builtin_amdgcn_s_memrealtime();
        cen[3] = (unsigned long long)g;
    }
    const int E0 = 2 + PW_MAX, F0 = 2 + 2 * PW_MAX;
    auto tileptr = [&](int r, int c) { return a.P + (size_t)r * LEAF + (size_t)c * LEAF * a.lda; };
    int idx[PW_MAX];
    if (g < W) {
        // ---- a diagonal strip: row g of the diagonal block, columns 0..g-1, then its own leaf.  Its solves need
        // E[c] only; the last column (c = g - 1) is the chain: solve -> update of (g,g) -> leaf without leaving the CU
        const bool dbg = PANEL_DBG && a.dbg && tid == 0;
        if (a.ver) {
            // task-queue driver: this strip's tiles have taken the updates of every earlier panel?
            const int vi = a.ver_i0 + (g >> 1);
            for (int c = 0; c <= g; ++c) idx[c] = (vi * a.ver_ld + a.ver_j0 + c) * (int)cholq::VS;
            if (!panel_wait_ge(a.ver, idx, g + 1, a.ver_need, a.dinfo, sh + 1, a.abort)) return false;
        }
        // Tiles (g, c'), 1 <= c' < g: whoever gets there first.  A helper workgroup that is running has put 1 into the
        // tile's word (3 once the tile carries every column c < c'); this strip takes the others for itself (2) and
        // updates them behind each of its solves as before -- it never waits for a workgroup that has not started.
        unsigned helped = 0;
        auto claim_tiles = [&]() {                      // (behind this strip's first solve: a helper that is going to run has started by now)
            if (tid == 0) {
                unsigned m = 0;
                for (int cp = 1; cp < g; ++cp) {
                    const int old = atomicCAS(a.flags + F0 + cp * PW_MAX + g, 0, 2);
                    if (old == 1 || old == 3) m |= 1u << cp;
                }
                sh[1] = (int)m;
            }
            __syncthreads();
            helped = (unsigned)sh[1];
            __syncthreads();
        };
        auto wait_helper = [&](int cp) -> bool {       // tile (g, cp) has taken the columns before cp?
            if (!((helped >> cp) & 1u)) return true;
            idx[0] = F0 + cp * PW_MAX + g;
            return panel_wait_ge(a.flags, idx, 1, 3, a.dinfo, sh + 1, a.abort);
        };
        for (int c = 0; c + 1 < g; ++c) {
            double *X = tileptr(g, c);
            XTile x;
            CTile cd;
            if (!wait_helper(c)) return false;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the updates this workgroup applied to (g,c) have
            __syncthreads();                                      // landed, whichever wave stored them
            trsm_prefetch(X, a.lda, x);
            diag_prefetch(tileptr(g, g), a.lda, cd);
            idx[0] = E0 + c;
            if (!panel_wait(a.flags, idx, 1, a.dinfo, sh + 1, a.abort)) return false;
            trsm_solve(s, tileptr(c, c), a.inv + (size_t)c * LEAF * LEAF, a.lda, x);
            trsm_store(X, a.lda, x);
            // X = L(g,c) goes to the strips below; the strip's own diagonal tile takes X X^T straight from the
            // accumulators (lower blocks only); the update of its tiles (g, c+1..g-1) needs L(c',c) of the diagonal
            // strips c' in (c, g)
            panel_publish_wt(a.flags + F0 + g * PW_MAX + c);
            diag_update_global(s, x, cd, tileptr(g, g), a.lda);
            if (c == 0 && a.NH > 0) claim_tiles();
            int cnt = 0;
            for (int cc = c + 1; cc < g; ++cc)
                if (!((helped >> cc) & 1u)) idx[cnt++] = F0 + cc * PW_MAX + c;
            if (cnt && !panel_wait(a.flags, idx, cnt, a.dinfo, sh + 1, a.abort)) return false;
            for (int t = 1; t < g - c; ++t)
                if (!((helped >> (c + t)) & 1u))
                    panel_product(s, -1.0, X, a.lda, tileptr(c + t, c), a.lda, 1.0, tileptr(g, c + t), a.lda);
        }
        if (g > 0) {
            const int c = g - 1;
            double *X = tileptr(g, c);
            XTile x;
            CTile cd;
            if (!wait_helper(c)) return false;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's updates of (g,c) and (g,g) have landed
            __syncthreads();
            trsm_prefetch(X, a.lda, x);
            diag_prefetch(tileptr(g, g), a.lda, cd);
            idx[0] = E0 + c;
            if (dbg) a.dbg[16 * g + 2] = __builtin_amdgcn_s_memrealtime();
            if (!panel_wait(a.flags, idx, 1, a.dinfo, sh + 1, a.abort)) return false;
            if (dbg) a.dbg[16 * g + 0] = __builtin_amdgcn_s_memrealtime();
            trsm_solve(s, tileptr(c, c), a.inv + (size_t)c * LEAF * LEAF, a.lda, x, a.dbg ? a.dbg + 16 * g + 6 : nullptr);
            trsm_store(X, a.lda, x);
            if (dbg) a.dbg[16 * g + 1] = __builtin_amdgcn_s_memrealtime();
            diag_update_into_leaf(s, x, cd, a.dbg ? a.dbg + 16 * g + 8 : nullptr);           // (ends behind a barrier that every wave's drained X stores precede)
            if (tid == 0) __hip_atomic_store((gint *)(a.flags + F0 + g * PW_MAX + c), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // this strip's diagonal tile has taken the updates of all earlier columns: factor + invert it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (dbg) a.dbg[16 * g + 4] = __builtin_amdgcn_s_memrealtime();
        leaf_body(s, (int)LEAF, tileptr(g, g), a.lda,
                  a.inv + (size_t)g * LEAF * LEAF, a.dinfo, a.goff + g * LEAF, (int)LEAF_FACTOR, nullptr,
                  a.flags + E0 + g, g > 0, dbg ? a.dbg + 16 * g + 11 : nullptr);
        panel_publish(a.flags + 2 + g);
        // task-queue driver: "something moved" (the workers drain their kernel instance when nothing does, cholq.h)
        if (a.abort && tid == 0)
            __hip_atomic_fetch_add((gint *)(a.abort + (cholq::Q_PROG - cholq::Q_ABORT)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (dbg) a.dbg[16 * g + 5] = __builtin_amdgcn_s_memrealtime();
        if (cen && tid == 0) cen[2] = __builtin_amdgcn_s_memrealtime();
        return true;
    }
    if (g < W + a.NH) {
        // ---- a helper: tile (gg, cp) of the diagonal block takes the columns c < cp as their rows X(gg, c), X(cp, c)
        // are published by the two diagonal strips (tickets 0 .. W-1: they started before this workgroup or are waited
        // for like the strips below wait for them)
        int h = g - W, cp = 1, gg = 2;
        while (h >= W - 1 - cp) { h -= W - 1 - cp; ++cp; }
        gg = cp + 1 + h;
        int *word = a.flags + F0 + cp * PW_MAX + gg;
        if (tid == 0) sh[1] = atomicCAS(word, 0, 1);
        __syncthreads();
        const int old = sh[1];
        __syncthreads();
        if (old != 0) return true;                      // strip gg was there first and does it itself
        if (a.ver) {
            idx[0] = ((a.ver_i0 + (gg >> 1)) * a.ver_ld + a.ver_j0 + cp) * (int)cholq::VS;
            if (!panel_wait_ge(a.ver, idx, 1, a.ver_need, a.dinfo, sh + 1, a.abort)) return false;
        }
        for (int c = 0; c < cp; ++c) {
            idx[0] = F0 + gg * PW_MAX + c;
            idx[1] = F0 + cp * PW_MAX + c;
            if (!panel_wait(a.flags, idx, 2, a.dinfo, sh + 1, a.abort)) return false;
            panel_product(s, -1.0, tileptr(gg, c), a.lda, tileptr(cp, c), a.lda, 1.0, tileptr(gg, cp), a.lda);
        }
        panel_publish_val(word, 3);
        return true;
    }
    const int gb = g - a.NH;
    if (a.tiles) {
        // ---- one tile (r, cp) of the rows below: left-looking -- it takes the columns c < cp as X(r, c) (its row's
        // progress word) and L(cp, c) (diagonal strip cp) appear, is solved on the early hand-off of leaf cp, and moves its
        // row's progress on.  Waits only for smaller tickets (same row, earlier column) and for diagonal strips.
        // Tiles are DRAWN in column-major order from a counter (the last word of the same spare block), so a tile's
        // predecessors were drawn before it by workgroups that are running or done -- whatever the number of workgroups.
        const int nbel = R - W, ntiles = nbel * W;
        int *const counter = a.flags + 2 * PFLAG_STRIDE - 1;
        for (;;) {
            if (tid == 0) sh[1] = atomicAdd(counter, 1);
            __syncthreads();
            const int q = sh[1];
            __syncthreads();
            if (q >= ntiles) return true;
            const int cp = q / nbel, r = W + q % nbel;
            const int pword = PFLAG_STRIDE + (r - W);
            if (a.ver) {
                idx[0] = ((a.ver_i0 + (r >> 1)) * a.ver_ld + a.ver_j0 + cp) * (int)cholq::VS;
                if (!panel_wait_ge(a.ver, idx, 1, a.ver_need, a.dinfo, sh + 1, a.abort)) return false;
            }
            for (int c = 0; c < cp; ++c) {
                idx[0] = pword;
                if (!panel_wait_ge(a.flags, idx, 1, c + 1, a.dinfo, sh + 1, a.abort)) return false;
                idx[0] = F0 + cp * PW_MAX + c;
                if (!panel_wait(a.flags, idx, 1, a.dinfo, sh + 1, a.abort)) return false;
                panel_product(s, -1.0, tileptr(r, c), a.lda, tileptr(cp, c), a.lda, 1.0, tileptr(r, cp), a.lda);
            }
            {
                double *X = tileptr(r, cp);
                XTile x;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                trsm_prefetch(X, a.lda, x);
                idx[0] = E0 + cp;
                if (!panel_wait(a.flags, idx, 1, a.dinfo, sh + 1, a.abort)) return false;
                trsm_solve(s, tileptr(cp, cp), a.inv + (size_t)cp * LEAF * LEAF, a.lda, x);
                trsm_store(X, a.lda, x);
            }
            // X went out write-through: drained stores, the barrier, then the row's progress word (as panel_publish_wt)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store((gint *)(a.flags + pword), cp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cp == W - 1) {
                if (a.tver) panel_publish_val(a.tver + (size_t)(a.tver_r0 + r) * cholq::VS, a.tver_val);
                if (a.abort && tid == 0)
                    __hip_atomic_fetch_add((gint *)(a.abort + (cholq::Q_PROG - cholq::Q_ABORT)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- strips below the diagonal block, shared round-robin by the other workgroups
    const int nw = G - W;
    const int r_first = W + (gb - W), r_step = nw > 0 ? nw : R;
    if (a.ver) {
        // task-queue driver: one strip per workgroup here; its W tiles carry the updates of every earlier panel?
        const int vi = a.ver_i0 + (r_first >> 1);
        for (int c = 0; c < W; ++c) idx[c] = (vi * a.ver_ld + a.ver_j0 + c) * (int)cholq::VS;
        if (r_first < R && !panel_wait_ge(a.ver, idx, W, a.ver_need, a.dinfo, sh + 1, a.abort)) return false;
    }
    const bool below_early = a.below_early != 0;
    for (int c = 0; c < W; ++c) {
        bool have_inv = false, have_early = false, have_rows = false;
        for (int r = r_first; r < R; r += r_step) {
            double *X = tileptr(r, c);
            if (c == W - 1 || below_early) {
                // the blocked solve, which can start on E[c], ~20 us before the leaf's full inverse is there (12 us
                // against the 20 of a staged product with the inverse); the updates of the later columns follow below
                XTile x;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                trsm_prefetch(X, a.lda, x);
                if (!have_early) {
                    idx[0] = E0 + c;
                    if (!panel_wait(a.flags, idx, 1, a.dinfo, sh + 1, a.abort)) return false;
                    have_early = true;
                }
                trsm_solve(s, tileptr(c, c), a.inv + (size_t)c * LEAF * LEAF, a.lda, x);
                trsm_store(X, a.lda, x);
                if (c == W - 1) continue;
            } else {
                if (!have_inv) {                           // inv(L_cc) published by strip c's workgroup
                    idx[0] = 2 + c;
                    if (!panel_wait(a.flags, idx, 1, a.dinfo, sh + 1, a.abort)) return false;
                    have_inv = true;
                }
                panel_product(s, 1.0, X, a.lda, a.inv + (size_t)c * LEAF * LEAF, (size_t)LEAF, 0.0, X, a.lda);
            }
            for (int t = 1; t <= W - 1 - c; ++t) {     // columns c+1..W-1 of this strip take the update
                if (t == 1 && !have_rows) {            // ... which needs L(c',c) of the diagonal strips c' in (c, W)
                    int cnt = 0;
                    for (int cc = c + 1; cc < W; ++cc) idx[cnt++] = F0 + cc * PW_MAX + c;
                    if (cnt && !panel_wait(a.flags, idx, cnt, a.dinfo, sh + 1, a.abort)) return false;
                    have_rows = true;
                }
                panel_product(s, -1.0, X, a.lda, tileptr(c + t, c), a.lda, 1.0, tileptr(r, c + t), a.lda);
            }
        }
    }
    if (a.tver && r_first < R) panel_publish_val(a.tver + (size_t)(a.tver_r0 + r_first) * cholq::VS, a.tver_val);
    if (a.abort && tid == 0)
        __hip_atomic_fetch_add((gint *)(a.abort + (cholq::Q_PROG - cholq::Q_ABORT)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cen && tid == 0) cen[2] = __builtin_amdgcn_s_memrealtime();
    return true;
}

__global__ __launch_bounds__(LT) void panel_kernel(const PanelArgs a)
{
    __shared__ double s[LEAF * ILD];   // the leaf's 128 x 130 block / the chain's 128 x 144 images; the products stage through its first 72 KiB
    __shared__ int sh[2];
    static_assert(2 * tile::BK * 2 * (LEAF + tile::PAD) <= LEAF_LDS && LEAF_LDS <= LEAF * ILD, "product staging and the leaf fit the buffer");
    const int tid = threadIdx.x;
    const int W = a.W, G = a.G;
    // Two tickets.  The diagonal strips -- the chain -- go to workgroups whose id is a multiple of 8, i.e. (with the
    // dispatcher's round-robin) to ONE XCD, so that what a leaf hands to the next strip is found in that XCD's L2
    // and not in memory; the strips below go to everybody else, and to the leftovers of the first kind.  Arrival
    // order within each kind: a workgroup still only waits for flags of workgroups that started before it or of
    // diagonal strips, which the first eligible workgroups to arrive take (the placement is a speed-only assumption).
    if (tid == 0) {
        int t = -1;
        if ((blockIdx.x & 7) == 0) {
            t = atomicAdd(a.flags, 1);
            if (t >= W) t = -1;
        }
        if (t < 0) t = W + atomicAdd(a.flags + 1, 1);
        sh[0] = t;
    }
    __syncthreads();
    const int g = sh[0];
    __syncthreads();
    if (g >= G + a.NH) return;
    (void)panel_strip(a, g, s, sh);
}

// The task-queue driver's form (cholq.h): ONE launch walks all panels of the block.  Workgroup b is strip b of every
// panel -- diagonal strip b while b < W, else row strip b - W of the next diagonal block -- and goes from one panel to the
// next without a kernel boundary: what it waits for are the version counters of its tiles.  (One launch per panel put 31
// command-processor dispatches on the path every worker spins on, and once in a few hundred factorisations one of them
// came 1.5 s late: tools/queue_stress.py.)
struct PanelSeq {
    double *A;              // origin of the block being factored
    size_t lda;
    int nblk;               // panels this kernel runs: 0 .. nblk-1 (the queue's part of the schedule)
    int nblk_all;           // panels of the whole schedule (the band of panel nblk-1 is the diagonal block of panel nblk, if any)
    const int *pstart;      // nblk_all + 1 panel boundaries (device)
    double *inv;            // leaf inverses of the block
    int *dinfo;
    int goff;               // global index of the block's first row (LAPACK info)
    int *flags;             // hand-off words of the block: panel starting at leaf column t at flags + t * PFLAG_STRIDE
    const int *ver;
    int ver_ld;
    int *tver;
    int *abort;
    unsigned long long *census;
    int helpers, tiles;     // the look-ahead driver's helper workgroups / tile-by-tile rows below (PanelArgs), from the surplus of the grid
};

__global__ __launch_bounds__(LT) void panel_seq_kernel(const PanelSeq q)
{
    __shared__ double s[LEAF * ILD];
    __shared__ int sh[2];
    const int g = blockIdx.x;
    for (int k = 0; k < q.nblk; ++k) {
        const int k0 = q.pstart[k], w = q.pstart[k + 1] - k0, wnext = k + 1 < q.nblk_all ? q.pstart[k + 2] - q.pstart[k + 1] : 0;
        PanelArgs a{};
        a.P = q.A + k0 + (size_t)k0 * q.lda; a.lda = q.lda;
        a.W = w / LEAF;
        a.R = a.G = (w + wnext) / LEAF;
        a.below_early = q.tiles;
        a.NH = (q.helpers && a.W > 2 && a.G + (a.W - 1) * (a.W - 2) / 2 <= (int)gridDim.x) ? (a.W - 1) * (a.W - 2) / 2 : 0;
        if (q.tiles && wnext > 0 && a.W >= 2 && (int)gridDim.x - a.W - a.NH >= 1) {
            a.tiles = 1;
            a.G = a.W + min((int)gridDim.x - a.W - a.NH, (a.R - a.W) * a.W);
        }
        a.inv = q.inv + (size_t)(k0 / LEAF) * LEAF * LEAF;
        a.dinfo = q.dinfo; a.goff = q.goff + k0;
        a.flags = q.flags + (size_t)(k0 / LEAF) * PFLAG_STRIDE;
        a.dbg = nullptr;
        a.ver = q.ver; a.ver_ld = q.ver_ld; a.ver_i0 = k0 / cholq::TM; a.ver_j0 = k0 / cholq::TN; a.ver_need = k0 / LEAF;
        a.tver = q.tver; a.tver_r0 = k0 / LEAF; a.tver_val = (k0 + w) / LEAF;
        a.abort = q.abort;
        a.census = (q.census && k < (int)cholq::TRACE_PANELS) ? q.census + 4 * cholq::TRACE_PANEL_WGS * (size_t)k : nullptr;
        if (g < a.G + a.NH && !panel_strip(a, g, s, sh)) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// Many independent factorisations of order <= PW_MAX leaves side by side in ONE launch (batch.hip: the population of a
// CMA-ES generation, python/05_tokamak/Split_SympGPR/main.py:63-66): each problem is a single panel whose W diagonal strips
// are W workgroups running the chain above on their own matrix.  Tickets in arrival order: ticket t is strip t % W of
// problem t / W, so a strip only ever waits for strips with smaller tickets -- resident or finished, whatever number of
// workgroups the chip holds at once.
struct PanelBatch {
    double *A;              // problem p at A + p * stride_a (column-major, lda), lower triangle filled
    size_t stride_a, lda;
    double *inv;            // leaf inverses of problem p at inv + p * stride_inv (W leaves)
    size_t stride_inv;
    int *flags;             // PFLAG_STRIDE ints per problem, zero on entry
    int *info;              // per problem: 0, the 1-based failing minor, or PANEL_TIMEOUT
    int *ticket;            // one int, zero on entry
    int W, nbatch;          // diagonal strips of this panel (its width / 128)
    int R;                  // row strips from the panel's first row to the end of the matrix (>= W): W .. R-1 are solved against it
    int k0;                 // first row / column of the panel inside its problem (a multiple of 128)
    int G;                  // workgroups per problem: W for the diagonal strips + G - W that share the R - W strips below
};

__global__ __launch_bounds__(LT) void panel_batch_kernel(const PanelBatch q)
{
    __shared__ double s[LEAF * ILD];
    __shared__ int sh[2];
    if (threadIdx.x == 0) sh[0] = atomicAdd(q.ticket, 1);
    __syncthreads();
    const int t = sh[0];
    __syncthreads();
    const int p = t / q.G, g = t - p * q.G;      // (diagonal strips 0 .. W-1 hold the smallest tickets of their problem)
    if (p >= q.nbatch) return;
    PanelArgs a{};
    a.P = q.A + (size_t)p * q.stride_a + (size_t)q.k0 + (size_t)q.k0 * q.lda; a.lda = q.lda;
    a.W = q.W; a.R = q.R; a.G = q.G;
    a.inv = q.inv + (size_t)p * q.stride_inv + (size_t)(q.k0 / (int)LEAF) * LEAF * LEAF;
    a.dinfo = q.info + p; a.goff = q.k0;
    a.flags = q.flags + (size_t)p * PFLAG_STRIDE;
    a.below_early = 1;            // (no helpers, no tiles here: a strip may only wait for smaller tickets, whatever the batch size)
    (void)panel_strip(a, g, s, sh);
}

inline int split(int n)
{
    // first part: a multiple of LEAF close to n/2 (>= LEAF, < n)
    int n1 = ((n / 2 + LEAF - 1) / LEAF) * LEAF;
    if (n1 >= n) n1 -= LEAF;
    return n1;
}

struct Ctx {
    double *inv;   // leaf inverses: leaf t at inv + t * LEAF * LEAF
    int *dinfo;
    hipStream_t st;
    int la_max = 0;   // potrf_rec hands blocks of order <= la_max to the look-ahead driver (0: never)
    int *flags = nullptr;   // hand-off flags of the panel kernel: PFLAG_STRIDE ints per leaf column, zeroed by potrf()
    void *qws = nullptr;    // the task-queue driver's part of the workspace (cholq.h), or null
};

int potrf_lookahead(int n, double *A, size_t lda, const Ctx &c, int nb, int off0);
// measured with the left-looking panel (factor ms at nb = 256 / 512 / 1024 / 2048): n = 8192 12.1 / 13.0 /
// 15.2 / 20.1, 16384 49.6 / 43.9 / 46.8 / 56.5, 32768 307 / 250 / 228 / 229, 49152 964 / 770 / 682 / 656
inline int la_block(int n) { return n <= 4096 ? 256 : (n <= 24576 ? 512 : (n <= 40960 ? 1024 : 2048)); }

int trsm_rec(int m, int n, const double *L, size_t ldl, double *B, size_t ldb, int off, const Ctx &c)
{
    if (n <= LEAF) {
        // B := B inv(L)^T, in place: one column tile (n <= 128), every workgroup owns full rows
        return gemm_nt(m, n, n, 1.0, B, ldb, c.inv + (size_t)(off / LEAF) * LEAF * LEAF, LEAF, 0.0,
                       B, ldb, 0, 0, c.st);
    }
    const int n1 = split(n), n2 = n - n1;
    int rc = trsm_rec(m, n1, L, ldl, B, ldb, off, c);
    if (rc) return rc;
    // B2 -= B1 L21^T
    rc = gemm_nt(m, n2, n1, -1.0, B, ldb, L + n1, ldl, 1.0, B + (size_t)n1 * ldb, ldb, 0, 0, c.st);
    if (rc) return rc;
    return trsm_rec(m, n2, L + n1 + (size_t)n1 * ldl, ldl, B + (size_t)n1 * ldb, ldb, off + n1, c);
}

// B (m x n) := B L^-1 (L lower, its leaf inverses in the workspace): the backward half of a
// multi-right-hand-side solve with the right-hand sides stored as ROWS.
//   X2 = B2 L22^-1 ;  B1 -= X2 L21 ("NN" product) ;  X1 = B1 L11^-1
int trsm_rl_rec(int m, int n, const double *L, size_t ldl, double *B, size_t ldb, int off, const Ctx &c)
{
    if (n <= LEAF)  // B := B inv(L), in place (one column tile, workgroups own full rows)
        return gemm_nn(m, n, n, 1.0, B, ldb, c.inv + (size_t)(off / LEAF) * LEAF * LEAF, LEAF, 0.0, B, ldb, c.st);
    const int n1 = split(n), n2 = n - n1;
    int rc = trsm_rl_rec(m, n2, L + n1 + (size_t)n1 * ldl, ldl, B + (size_t)n1 * ldb, ldb, off + n1, c);
    if (rc) return rc;
    rc = gemm_nn(m, n1, n2, -1.0, B + (size_t)n1 * ldb, ldb, L + n1, ldl, 1.0, B, ldb, c.st);
    if (rc) return rc;
    return trsm_rl_rec(m, n1, L, ldl, B, ldb, off, c);
}

int potrf_rec(int n, double *A, size_t lda, int off, const Ctx &c)
{
    if (n <= LEAF) {
        hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(LT), 0, c.st, n, A, lda,
                           c.inv + (size_t)(off / LEAF) * LEAF * LEAF, c.dinfo, off, (int)LEAF_FACTOR,
                           (unsigned long long *)nullptr);
        SGPR_CHECK_LAUNCH();
        return 0;
    }
    // mid-size blocks (also the halves a large recursive factorisation splits into): blocked
    // right-looking with the next panel on a side stream
    if (n > 4 * LEAF && n <= c.la_max) return potrf_lookahead(n, A, lda, c, 0, off);
    const int n1 = split(n), n2 = n - n1;
    double *A21 = A + n1, *A22 = A + n1 + (size_t)n1 * lda;
    int rc = potrf_rec(n1, A, lda, off, c);
    if (rc) return rc;
    rc = trsm_rec(n2, n1, A, lda, A21, lda, off, c);
    if (rc) return rc;
    rc = gemm_nt(n2, n2, n1, -1.0, A21, lda, A21, lda, 1.0, A22, lda, 1, 0, c.st);  // SYRK, lower
    if (rc) return rc;
    return potrf_rec(n2, A22, lda, off + n1, c);
}

int trsv_n_rec(int n, const double *L, size_t ldl, double *b, int off, const Ctx &c)
{
    if (n <= LEAF) {
        hipLaunchKernelGGL(leaf_matvec_kernel, dim3(1), dim3(MV_T), 0, c.st, n,
                           c.inv + (size_t)(off / LEAF) * LEAF * LEAF, b, 0);
        SGPR_CHECK_LAUNCH();
        return 0;
    }
    if (n <= TRSV_BLOCK) {
        hipLaunchKernelGGL(trsv_block_kernel, dim3(1), dim3(MV_T), 0, c.st, n, L, ldl,
                           c.inv + (size_t)(off / LEAF) * LEAF * LEAF, b, 0);
        SGPR_CHECK_LAUNCH();
        return 0;
    }
    const int n1 = split(n), n2 = n - n1;
    int rc = trsv_n_rec(n1, L, ldl, b, off, c);
    if (rc) return rc;
    rc = gemv_n_sub(n2, n1, L + n1, ldl, b, b + n1, c.st);  // b2 -= L21 b1
    if (rc) return rc;
    return trsv_n_rec(n2, L + n1 + (size_t)n1 * ldl, ldl, b + n1, off + n1, c);
}

int trsv_t_rec(int n, const double *L, size_t ldl, double *b, int off, const Ctx &c)
{
    if (n <= LEAF) {
        hipLaunchKernelGGL(leaf_matvec_kernel, dim3(1), dim3(MV_T), 0, c.st, n,
                           c.inv + (size_t)(off / LEAF) * LEAF * LEAF, b, 1);
        SGPR_CHECK_LAUNCH();
        return 0;
    }
    if (n <= TRSV_BLOCK) {
        hipLaunchKernelGGL(trsv_block_kernel, dim3(1), dim3(MV_T), 0, c.st, n, L, ldl,
                           c.inv + (size_t)(off / LEAF) * LEAF * LEAF, b, 1);
        SGPR_CHECK_LAUNCH();
        return 0;
    }
    const int n1 = split(n), n2 = n - n1;
    int rc = trsv_t_rec(n2, L + n1 + (size_t)n1 * ldl, ldl, b + n1, off + n1, c);
    if (rc) return rc;
    rc = gemv_t_sub(n2, n1, L + n1, ldl, b + n1, b, c.st);  // b1 -= L21^T b2
    if (rc) return rc;
    return trsv_t_rec(n1, L, ldl, b, off, c);
}

int leaves_invert_only(int n, double *L, size_t ldl, const Ctx &c)
{
    for (int off = 0; off < n; off += LEAF) {
        const int nb = n - off < LEAF ? n - off : LEAF;
        hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(LT), 0, c.st, nb, L + off + (size_t)off * ldl, ldl,
                           c.inv + (size_t)(off / LEAF) * LEAF * LEAF, c.dinfo, off,
                           (int)LEAF_INVERT_ONLY, (unsigned long long *)nullptr);
        SGPR_CHECK_LAUNCH();
    }
    return 0;
}

inline size_t inv_bytes(int n)
{
    return (size_t)((n + LEAF - 1) / LEAF) * LEAF * LEAF * sizeof(double);
}
// flag block of the panel that starts at leaf column t: flags + t * PFLAG_STRIDE
inline size_t flag_bytes(int n)
{
    return (((size_t)((n + LEAF - 1) / LEAF) * PFLAG_STRIDE * sizeof(int)) + 255) / 256 * 256;
}


}  // namespace

// [ leaf inverses | hand-off words | two n-vectors the strip solves publish their segments through ]
inline size_t pub_bytes(int n) { return ((size_t)2 * n * sizeof(double) + 255) / 256 * 256; }
size_t potrf_workspace(int n) { return n <= 0 ? 256 : inv_bytes(n) + flag_bytes(n) + pub_bytes(n) + cholq::ws_bytes(n) + 256; }

namespace {

// Right-looking blocked factorisation with one panel of look-ahead on a side stream -- the form
// used below ~50k, where the recursion's serial chain of leaves and small panel solves would leave
// most of the chip idle.  Per block column k (width nb):
//     P stream:  factor A(k,k) (recursively, leaves in LDS), solve the panel below it
//     U stream:  U1 = update of block column k+1 only  ->  signals P to start panel k+1
//                U2 = the rest of the trailing update (the big MFMA SYRK), runs beside panel k+1
// Measured (factor stage, ms, recursive -> this): n = 8192 21.9 -> 16.4, 16384 66 -> 53,
// 32768 270 -> 240, 49152 766 -> 705, 65536 1537 -> 1578 (so the recursion takes over there).
// The overlap is partial: every CU is held by a workgroup of the big update for ~100 us at a time,
// so the panel's short dependent kernels queue behind them (they run 2x longer than alone).
// Reserving CUs for the panel with a CU-masked stream was tried and is slower overall (the update
// loses 6 % of the chip and the tile map its 256-CU geometry).
// The numbers are the same operations in a different order; the leaf workspace layout is shared with
// the recursive driver, so the solves do not care which one produced L.
// Events of one factorisation, BORROWED from a pool the calling thread keeps per device and never gives back.
// Creating and destroying them per call (rounds 1-2) put runtime housekeeping -- event / signal memory being released while
// the device is still running the factorisation that used it -- beside kernels whose workgroups wait for each other; see
// DESIGN 3.9 for what that did to the task-queue driver.  Re-recording an event later is safe: a stream's wait refers to
// the record that was current when the wait was enqueued.  Nested sets (the queue driver, then the look-ahead driver for
// the block it leaves) take consecutive slices of the pool.
struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t top = 0;
};
thread_local EventPool t_event_pool[64];

struct EventSet {
    std::vector<hipEvent_t> ev;
    EventPool *pool = nullptr;
    size_t base = 0;
    int create(size_t count)
    {
        int dev = 0;
        SGPR_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) { set_error("potrf: device index out of range"); return SGPR_E_ARG; }
        pool = &t_event_pool[dev];
        base = pool->top;
        while (pool->ev.size() < base + count) {
            hipEvent_t e = nullptr;
            SGPR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            pool->ev.push_back(e);
        }
        pool->top = base + count;
        ev.assign(pool->ev.begin() + (long)base, pool->ev.begin() + (long)(base + count));
        return 0;
    }
    ~EventSet()
    {
        if (pool) pool->top = base;
    }
};
// whatever happens in between, the caller's stream waits for everything queued on the side stream
struct StreamJoin {
    hipStream_t side, caller;
    hipEvent_t ev;
    ~StreamJoin()
    {
        if (hipEventRecord(ev, side) != hipSuccess || hipStreamWaitEvent(caller, ev, 0) != hipSuccess)
            (void)hipStreamSynchronize(side);
        gemm_set_overlap(0);
    }
};

// The high-priority side stream of the device that owns the caller's stream (created on first use, ONE per device,
// shared by every handle and every host thread; release_device() gives it back).
//
// Forward progress with several handles factoring at once: every panel_kernel of a device is launched on this one
// in-order stream, so at most one panel kernel -- one set of spinning strips -- is resident per device at any time,
// whatever number of handles / threads are factoring.  The workgroups it waits for are its own (dispatched in block
// order, each as soon as ONE CU drains); everything else on the device is an MFMA / Gram / solve launch whose
// workgroups end by themselves.  tests/test_gpu_threads.py runs three handles at n = 16384 against this.
hipStream_t g_side[64] = {};
std::mutex g_side_mu;                                      // two fit handles may factor for the first time at once

// The streams this library creates are handed back by an exit handler of the LIBRARY, registered when the first of them is
// created: atexit handlers run in reverse order of registration, the HIP runtime registered its own when it was initialised
// (before any stream could exist), so this one runs while the runtime is still whole.  Round 3 did this from the Python
// binding only (a run under rocprofv3 --kernel-trace that had used the CU-masked streams crashed in an exit handler when
// they were left to the runtime's own teardown); users of the C ABI get the same now.  No synchronisation here: a stream
// that is still busy is released when its work ends.
void release_all_streams_at_exit();
void register_exit_release()
{
    static std::once_flag once;
    std::call_once(once, [] { (void)atexit(release_all_streams_at_exit); });
}

int side_stream(hipStream_t caller, hipStream_t *out, int *dev_out)
{
    hipStream_t *const side = g_side;
    int dev = -1;
    if (caller) {
        SGPR_HIP(hipStreamGetDevice(caller, &dev));        // the device that owns the caller's stream
    } else {
        SGPR_HIP(hipGetDevice(&dev));                      // the null stream belongs to the current device
    }
    if (dev < 0 || dev >= 64) { set_error("potrf: device index out of range"); return SGPR_E_ARG; }
    std::lock_guard<std::mutex> lock(g_side_mu);
    if (!side[dev]) {
        int cur = -1, lo = 0, hi = 0;
        SGPR_HIP(hipGetDevice(&cur));
        if (cur != dev) SGPR_HIP(hipSetDevice(dev));
        hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&side[dev], hipStreamNonBlocking, hi);
        if (e == hipSuccess) register_exit_release();
        if (cur != dev) (void)hipSetDevice(cur);
        SGPR_HIP(e);
    }
    *out = side[dev];
    *dev_out = dev;
    return 0;
}

// ---- task-queue driver (cholq.h) ----------------------------------------------------------------------------------
// One persistent worker grid runs every trailing-update tile and every rows-below solve of the block from an ordered
// task list; the chain of diagonal blocks runs as one panel kernel per panel (diagonal strips + the rows of the next
// diagonal block) beside it, each strip starting as soon as its own tiles carry the updates of all earlier panels
// (version counters, no kernel boundary).
//
// A panel workgroup (147 KiB of LDS) cannot share a CU with a worker (104 KiB), so the two kernels run on DISJOINT CU
// SETS: two streams with CU masks (hipExtStreamCreateWithCUMask), R CUs of every XCD for the panel kernels, the rest for
// the workers.  Merely launching fewer workers than CUs is not enough: the dispatcher binds a workgroup to an XCD and a
// shader engine round-robin BEFORE it looks for a free CU, and a panel workgroup bound to an engine that the persistent
// workers fill waits there for good, free CUs next door or not (tools/queue_fail_hunt.py caught it: one panel workgroup
// started 1.09 s late, on the first CU a worker gave back).  With the masks an engine without a CU of the queue's set is
// never chosen.  (Mask bit i is CU i / 8 of XCD i % 8: tools/probe_cumask.py.)
// Queue factorisations of one device are serialised: they share these two streams and their CU sets.
struct QueueDevice {
    std::mutex mu;
    hipEvent_t done = nullptr;                 // end of the last queue factorisation enqueued on this device
    int ncu = 0;
    hipStream_t workers[4] = {}, panels[4] = {};   // by R - 1 (CUs per XCD set aside for the panel kernels)
    bool overlap_ok[4] = {};                   // ... and whether kernels on the pair have been SEEN to run side by side
    bool failed = false;                       // masked streams cannot be had, kernels on them do not overlap, or a queue
                                               // factorisation gave up: the look-ahead driver takes over (until the streams are released)
};
QueueDevice g_qdev[64];

// Two one-thread kernels, one on each stream of the pair, that wait for each other (bounded: 50 ms).  The task-queue
// factorisation is two persistent kernels that hand work to each other: where dispatches are serialised -- a profiler collecting
// counters, HIP_LAUNCH_BLOCKING, AMD_SERIALIZE_KERNEL, a debugger -- the second never starts while the first waits, and the
// factorisation would end in its time limit with A half overwritten.  Tried once per stream pair; w: 4 ints, zero.
__global__ void overlap_probe_kernel(int *w, int me)
{
    __hip_atomic_store((gint *)(w + me), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int seen = 0;
    while (!(seen = __hip_atomic_load((gint *)(w + 1 - me), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __builtin_amdgcn_s_sleep(32);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) break;
    }
    w[2 + me] = seen ? 1 : 2;
}
static bool queue_overlap_ok(hipStream_t sw, hipStream_t sp)
{
    int *w = nullptr;
    if (hipMalloc((void **)&w, 4 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return false; }
    int h[4] = {0, 0, 0, 0};
    bool ok = hipMemset(w, 0, sizeof(h)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(overlap_probe_kernel, dim3(1), dim3(1), 0, sp, w, 0);
        hipLaunchKernelGGL(overlap_probe_kernel, dim3(1), dim3(1), 0, sw, w, 1);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(sp) == hipSuccess && hipStreamSynchronize(sw) == hipSuccess &&
             hipMemcpy(h, w, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[2] == 1 && h[3] == 1;
    }
    (void)hipFree(w);
    (void)hipGetLastError();
    return ok;
}

int queue_streams(QueueDevice &qd, int dev, int R, hipStream_t *sw, hipStream_t *sp)
{
    if (!qd.workers[R - 1]) {
        int cur = -1;
        SGPR_HIP(hipGetDevice(&cur));
        if (cur != dev) SGPR_HIP(hipSetDevice(dev));
        const int words = qd.ncu / 32;
        std::vector<uint32_t> mp((size_t)words, 0u), mw((size_t)words, 0xFFFFFFFFu);
        for (int b = 0; b < 8 * R; ++b) { mp[b / 32] |= 1u << (b % 32); mw[b / 32] &= ~(1u << (b % 32)); }
        hipError_t e = hipExtStreamCreateWithCUMask(&qd.panels[R - 1], (uint32_t)words, mp.data());
        if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&qd.workers[R - 1], (uint32_t)words, mw.data());
        if (e == hipSuccess) register_exit_release();
        if (cur != dev) (void)hipSetDevice(cur);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            qd.failed = true;
            return SGPR_E_HIP;
        }
    }
    if (!qd.overlap_ok[R - 1]) {
        if (!queue_overlap_ok(qd.workers[R - 1], qd.panels[R - 1])) {
            qd.failed = true;
            return SGPR_E_HIP;
        }
        qd.overlap_ok[R - 1] = true;
    }
    *sw = qd.workers[R - 1];
    *sp = qd.panels[R - 1];
    return 0;
}

thread_local bool t_last_potrf_used_queue = false;   // set by potrf_queue once its kernels are enqueued, cleared by every potrf()

// returns 1 when the queue form cannot be used here (the caller falls back to the look-ahead driver), < 0 on errors
int potrf_queue(int n, double *A, size_t lda, const Ctx &c, int off0)
{
    int dev = -1;
    if (c.st) {
        SGPR_HIP(hipStreamGetDevice(c.st, &dev));
    } else {
        SGPR_HIP(hipGetDevice(&dev));
    }
    if (dev < 0 || dev >= 64) { set_error("potrf: device index out of range"); return SGPR_E_ARG; }
    QueueDevice &qd = g_qdev[dev];
    std::lock_guard<std::mutex> lock(qd.mu);
    if (qd.failed) return 1;
    if (!qd.ncu) {
        SGPR_HIP(hipDeviceGetAttribute(&qd.ncu, hipDeviceAttributeMultiprocessorCount, dev));
        SGPR_HIP(hipEventCreateWithFlags(&qd.done, hipEventDisableTiming));
        SGPR_HIP(hipEventRecord(qd.done, c.st));
    }
    if (qd.ncu % 64 != 0 || qd.ncu < 64) { qd.failed = true; return 1; }      // 8 XCDs, whole mask words
    const std::vector<int> starts = cholq::default_starts(n);
    int band = 0;       // workgroups of the widest panel kernel: its diagonal strips + the rows of the next diagonal block
    for (size_t k = 0; k + 1 < starts.size(); ++k)
        band = std::max(band, (starts[k + 1] - starts[k] + (k + 2 < starts.size() ? starts[k + 2] - starts[k + 1] : 0)) / LEAF);
    // helper workgroups for the tiles inside the diagonal blocks and extra ones for the rows of the next diagonal block tile
    // by tile (tunables "q_helpers", "q_tiles" = number of extra workgroups, 0 = strips as before)
    static const int q_helpers = (int)tune("q_helpers", 0), q_tiles = (int)tune("q_tiles", 0);   // measured at n = 16384: 29.6 ms without, 30.1 - 30.8 with (8 workers fewer)
    int nh_max = 0;
    for (size_t k = 0; k + 1 < starts.size(); ++k) {
        const int W = (starts[k + 1] - starts[k]) / LEAF;
        if (q_helpers && W > 2) nh_max = std::max(nh_max, (W - 1) * (W - 2) / 2);
    }
    const int pgrid = band + nh_max + std::max(0, q_tiles);
    const int R = (pgrid + 7) / 8;
    if (R > 4) return 1;
    hipStream_t sw = nullptr, sp = nullptr;
    if (queue_streams(qd, dev, R, &sw, &sp)) return 1;
    // tunable "q_slack" = <CUs of the worker set left empty> (default 0; sgpr_probe_tune).  32 = one per shader engine: with that much room the
    // workgroups of a preempted-and-restored grid all find a CU again (the stall) -- at 13 % of the throughput.
    static const int qslack = (int)tune("q_slack", 0);
    const int nworkers = qd.ncu - 8 * R - std::max(0, std::min(qslack, 64));
    const cholq::Plan *plan = cholq::get_plan(n, nworkers);
    if (!plan) return 1;
    const cholq::Ws ws = cholq::carve(c.qws, n);
    const hipStream_t su = c.st;
    int rc;
    SGPR_HIP(hipStreamWaitEvent(su, qd.done, 0));               // after the previous queue factorisation on this device
    if ((rc = cholq::prepare(*plan, ws, su))) return rc;
    if (cholq::forced_giveup()) {                                // tests only: the give-up word is up, and so is what a timed-out waiter leaves
        static const int k_timeout = POTRF_HANDOFF_TIMEOUT;
        SGPR_HIP(hipMemcpyAsync(c.dinfo, &k_timeout, sizeof(int), hipMemcpyHostToDevice, su));
    }
    EventSet es;
    if ((rc = es.create(3))) return rc;
    const int tn = n / cholq::TN;
    double *inv_blk = c.inv + (size_t)(off0 / LEAF) * LEAF * LEAF;
    int *flags_blk = c.flags + (size_t)(off0 / LEAF) * PFLAG_STRIDE;
    {
        SGPR_HIP(hipEventRecord(es.ev[0], su));                 // both queue streams join the caller's stream ...
        SGPR_HIP(hipStreamWaitEvent(sp, es.ev[0], 0));
        SGPR_HIP(hipStreamWaitEvent(sw, es.ev[0], 0));
        StreamJoin join_p{sp, su, es.ev[1]};                    // ... and leave it again on every exit path
        StreamJoin join_w{sw, su, es.ev[2]};
        PanelSeq ps{};
        ps.A = A; ps.lda = lda; ps.nblk = plan->nq; ps.nblk_all = plan->nblk; ps.pstart = ws.pstart;
        ps.inv = inv_blk; ps.dinfo = c.dinfo; ps.goff = off0; ps.flags = flags_blk;
        ps.ver = ws.ver; ps.ver_ld = tn; ps.tver = ws.tver; ps.abort = cholq::abort_word(ws);
        ps.census = cholq::trace_panel_base((int)(plan->tasks.size() / 2));
        ps.helpers = q_helpers != 0; ps.tiles = q_tiles > 0;
        t_last_potrf_used_queue = true;
        hipLaunchKernelGGL(panel_seq_kernel, dim3(pgrid), dim3(LT), 0, sp, ps);
        SGPR_CHECK_LAUNCH();
        if ((rc = cholq::launch_workers(*plan, ws, A, lda, inv_blk, flags_blk, c.dinfo, PFLAG_STRIDE, sw))) return rc;
    }
    SGPR_HIP(hipEventRecord(qd.done, su));
    cholq::remember(ws, n, (int)(plan->tasks.size() / 2));
    cholq::remember_plan(plan);          // for the post-mortem probe
    static const bool qdebug = tune("q_debug", 0) != 0;
    if (qdebug) {
        SGPR_HIP(hipStreamSynchronize(su));
        cholq::postmortem(false);
    }
    if (plan->nq < plan->nblk) {
        // the block the queue left (every update of its panels applied, its rows of L final): the look-ahead driver
        const int S = plan->starts[plan->nq];
        Ctx ct = c;
        ct.qws = nullptr;
        // The host WAITS here.  With the look-ahead driver's ~100 launches enqueued behind the persistent kernels (on this
        // stream and on the high-priority side stream, all of them blocked by the joins above) the stall of DESIGN 3.9 was no
        // longer rare but came in the first factorisation, every time, ~4 ms in; with nothing enqueued behind them 400 in a
        // row were clean (and 1 in ~300 - 1700 still stalls, as with the whole factorisation in the queue).
        static const bool nosync = tune("q_nosync", 0) != 0;               // tests of the drain-and-relaunch recovery only
        if (!nosync) SGPR_HIP(hipStreamSynchronize(su));
        return potrf_lookahead(n - S, A + S + (size_t)S * lda, lda, ct, 0, off0 + S);
    }
    return 0;
}

void release_all_streams_at_exit()
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return; }
    for (int dev = 0; dev < ndev && dev < 64; ++dev) {
        hipStream_t drop[9] = {};
        int nd = 0;
        {
            std::lock_guard<std::mutex> lock(g_side_mu);
            if (g_side[dev]) { drop[nd++] = g_side[dev]; g_side[dev] = nullptr; }
        }
        {
            QueueDevice &qd = g_qdev[dev];
            std::lock_guard<std::mutex> lock(qd.mu);
            for (int r = 0; r < 4; ++r) {
                if (qd.workers[r]) { drop[nd++] = qd.workers[r]; qd.workers[r] = nullptr; }
                if (qd.panels[r]) { drop[nd++] = qd.panels[r]; qd.panels[r] = nullptr; }
            }
            qd.failed = true;                  // nothing new is started on this device from here on
        }
        if (!nd) continue;
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != dev) (void)hipSetDevice(dev);
        for (int q = 0; q < nd; ++q) (void)hipStreamDestroy(drop[q]);
        if (cur != dev && cur >= 0) (void)hipSetDevice(cur);
    }
    (void)hipGetLastError();
}

int release_streams(int dev)
{
    // drain and destroy the streams this library created on `dev` (side stream, the queue driver's masked pair and its
    // event); the next factorisation creates them again.  The caller guarantees no factorisation is being enqueued.
    {
        // nothing of ours on this device: do not even touch it (an exit hook calls this for every device of the box)
        bool any = false;
        {
            std::lock_guard<std::mutex> lock(g_side_mu);
            any = g_side[dev] != nullptr;
        }
        QueueDevice &qd = g_qdev[dev];
        std::lock_guard<std::mutex> lock(qd.mu);
        for (int r = 0; r < 4; ++r) any = any || qd.workers[r] || qd.panels[r];
        any = any || qd.done;
        if (!any) return 0;
    }
    int cur = -1;
    SGPR_HIP(hipGetDevice(&cur));
    if (cur != dev) SGPR_HIP(hipSetDevice(dev));
    hipError_t first = hipSuccess;
    auto drop = [&](hipStream_t &s) {
        if (!s) return;
        hipError_t e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipStreamDestroy(s);
        if (e != hipSuccess && first == hipSuccess) first = e;
        s = nullptr;
    };
    {
        std::lock_guard<std::mutex> lock(g_side_mu);
        drop(g_side[dev]);
    }
    {
        QueueDevice &qd = g_qdev[dev];
        std::lock_guard<std::mutex> lock(qd.mu);
        for (int r = 0; r < 4; ++r) { drop(qd.workers[r]); drop(qd.panels[r]); }
        if (qd.done) {
            const hipError_t e = hipEventDestroy(qd.done);
            if (e != hipSuccess && first == hipSuccess) first = e;
            qd.done = nullptr;
        }
        qd.ncu = 0;
        qd.failed = false;
        for (int r = 0; r < 4; ++r) qd.overlap_ok[r] = false;
    }
    if (cur != dev) (void)hipSetDevice(cur);
    SGPR_HIP(first);
    return 0;
}

int potrf_lookahead(int n, double *A, size_t lda, const Ctx &c, int nb, int off0)
{
    if (nb == 0 && c.qws && c.flags && cholq::eligible(n) && (lda & 1) == 0 && (((uintptr_t)A & 15) == 0) && off0 % LEAF == 0) {
        const int rq = potrf_queue(n, A, lda, c, off0);
        if (rq <= 0) return rq;                 // done, or a real error; 1: not available here
    }
    hipStream_t sp = nullptr;
    {
        int dev = -1;
        const int rc0 = side_stream(c.st, &sp, &dev);
        if (rc0) return rc0;
    }
    // Panel kernel usable?  (The multi-launch panels below remain for shapes it does not take and for A/B runs.)
    static const int pmode = (int)tune("la_panel", 3);     // 3: the persistent panel kernel; 1 left-looking, 2 recursive, 0 column-wise multi-launch panels
    const bool fused_cap = pmode == 3 && c.flags && n % LEAF == 0 && (lda & 1) == 0 && (((uintptr_t)A & 15) == 0) &&
                           off0 % LEAF == 0;
    // Block schedule.  nb > 0: uniform width (the caller's choice).  nb == 0: widths follow the order of
    // what is left: wide blocks (alone, k = 2048 / 1024 / 512 products run at 61 / 57 / 49 TFLOP/s) while the trailing
    // update is what each step waits for, narrower ones once the chain of leaves is (the chain costs the
    // same per column at any width, and a narrow step loses less to its own update of block column k+1).
    std::vector<int> starts;
    {
        // 2048-wide steps (k = 2048 updates: 61 TFLOP/s) pay off only when the early updates are long enough to
        // hide a 16-leaf chain: measured n = 32768 198.5 -> 194.0 ms, 49152 620 -> 603, but 16384 34.0 -> 35.6
        static const int t0_env = (int)tune("la_t0", 0);
        const int t0 = t0_env > 0 ? t0_env : (n >= 24576 ? 12288 : 1 << 30);
        static const int t1 = (int)tune("la_t1", 6144);   // swept 2048 .. 9216: n = 12288 18.7 -> 18.1 ms, 16384 34.2 -> 33.4
        static const int t2 = (int)tune("la_t2", 2048);
        for (int pos = 0; pos < n;) {
            starts.push_back(pos);
            const int rem = n - pos;
            int w = nb > 0 ? nb : (rem > t0 ? 2048 : (rem > t1 ? 1024 : (rem > t2 ? 512 : 256)));
            if (nb == 0 && !fused_cap) w = la_block(n);
            pos += std::min(w, rem);
        }
        starts.push_back(n);
    }
    const int nblk = (int)starts.size() - 1;
    int wmax = 0;
    for (int k = 0; k < nblk; ++k) wmax = std::max(wmax, starts[k + 1] - starts[k]);
    EventSet es;
    {
        const int rc0 = es.create(3 * (size_t)nblk + 3);
        if (rc0) return rc0;
    }
    std::vector<hipEvent_t> &ev = es.ev;
    const hipStream_t su = c.st;
    Ctx cp{c.inv, c.dinfo, sp, 0, c.flags};
    // Panel k = diagonal block + everything below it.  Default: ONE launch of the persistent panel kernel
    // (above).  The multi-launch forms: SGPR_LA_PANEL=c "column": leaf column by leaf column, factor the
    // 128 x 128 leaf, multiply the rows below by its inverse, fold that column into the panel's remaining
    // columns (3 launches per leaf column; round 1's default); =l folds left-looking instead; =r factors
    // the diagonal block recursively and then solves the rows below (19 launches for nb = 512).
    const bool fused_ok = fused_cap && wmax % LEAF == 0 && wmax / LEAF <= PW_MAX;
    const bool left = pmode != 2, right_in = pmode == 0 || pmode == 3;
    auto panel = [&](int k) -> int {   // on the P stream
        const int k0 = starts[k], w = starts[k + 1] - k0, rows = n - k0, below = rows - w;
        double *Akk = A + k0 + (size_t)k0 * lda;
        if (fused_ok) {
            PanelArgs pa{};
            pa.P = Akk; pa.lda = lda;
            pa.R = rows / LEAF; pa.W = w / LEAF;
            const int nbelow = pa.R - pa.W;
            // While the step is bound by the trailing update running beside the panel (not by the chain of
            // leaves), the kernel takes the diagonal block only -- W workgroups, the chain -- and the rows
            // below are solved by the grid-wide MFMA kernel afterwards: inside the persistent kernel a strip
            // below costs ~W (W + 3) / 2 products of 128^3, each on a CU of its own at 20 us alone and 2-3x
            // that beside a bandwidth-hungry update (a 1024-wide panel over 14336 rows: 3.1 - 5.7 ms).
            // Once the chain is what the step waits for, every strip gets its own workgroup: one launch, the
            // strips below finish ~20 us after the last leaf.
            const double m_u2 = (double)(rows + w);                    // order of the update running beside
            const double u2_us = k > 0 ? m_u2 * m_u2 * (starts[k] - starts[k - 1]) / 55e6 : 0.0;
            static const double split_ratio = tune("panel_split_ratio", 3.0);
            const bool split = nbelow > 0 && k > 0 && u2_us > split_ratio * (125.0 * pa.W + 100.0);
            if (split) {
                pa.R = pa.W;
                pa.G = pa.W;
            } else {
                // strips below per workgroup: as many as still finish under the update (each workgroup holds
                // a whole CU, 133 KiB of LDS, that the update loses); one each once the chain is the limit
                const double strip_us = 30.0 * (pa.W * (pa.W + 1) / 2 + pa.W), chain_us = 125.0 * pa.W;
                int per_wg = (int)((0.7 * u2_us - 0.5 * chain_us) / strip_us);
                per_wg = std::max(1, std::min(per_wg, 4));
                const int nwg = std::min((nbelow + per_wg - 1) / per_wg, PANEL_G_MAX - pa.W);
                pa.G = pa.W + (nbelow > 0 ? std::max(1, nwg) : 0);
            }
            // helpers for the tiles inside the diagonal block (tunable "panel_helpers", default on) and the early-hand-off
            // solve for every column of the strips below ("panel_below_early")
            static const bool helpers_on = tune("panel_helpers", 1) != 0, below_early_on = tune("panel_below_early", 1) != 0;
            pa.NH = helpers_on && pa.W > 2 ? (pa.W - 1) * (pa.W - 2) / 2 : 0;
            if (pa.G + pa.NH > PANEL_G_MAX) pa.NH = 0;
            pa.below_early = below_early_on ? 1 : 0;
            // one workgroup per tile of the rows below while they all fit the chip (the chain-bound sizes: a strip below is
            // W solves + W (W - 1) / 2 products of 20 - 30 us on ONE CU, 265 us for W = 4 against the chain's 215)
            static const bool tiles_on = tune("panel_tiles", 1) != 0;
            static const double tile_mult = tune("panel_tile_mult", 1.0);   // 1 / 2 / 3: n = 8192 6.94 / 7.06 / 7.12 ms, 12288 16.1 / 16.5 / 16.6
            if (tiles_on && !split && nbelow > 0 && pa.W >= 2 && nbelow < PFLAG_STRIDE) {
                // workgroups for the tiles: all of them while the step waits for the chain anyway, else a multiple of the
                // strips' number (every one holds a CU that the update beside loses)
                const double chain_us = 60.0 * pa.W + 50.0;
                int nt = u2_us < chain_us ? nbelow * pa.W : (int)(tile_mult * (pa.G - pa.W));
                nt = std::max(1, std::min(std::min(nt, nbelow * pa.W), PANEL_G_MAX - pa.W - pa.NH));
                pa.tiles = 1;
                pa.G = pa.W + nt;
            }
            const int t0 = (off0 + k0) / LEAF;
            pa.inv = cp.inv + (size_t)t0 * LEAF * LEAF;
            pa.dinfo = cp.dinfo; pa.goff = off0 + k0;
            pa.flags = c.flags + (size_t)t0 * PFLAG_STRIDE;
            pa.dbg = (PANEL_DBG && g_panel_dbg) ? g_panel_dbg + (size_t)16 * t0 : nullptr;
            // (at least W workgroups with an id that is a multiple of 8; surplus ones find no ticket and leave)
            hipLaunchKernelGGL(panel_kernel, dim3(std::max(pa.G + pa.NH, 8 * (pa.W - 1) + 1)), dim3(LT), 0, sp, pa);
            SGPR_CHECK_LAUNCH();
            if (split) return trsm_rec(below, w, Akk, lda, Akk + w, lda, off0 + k0, cp);
            return 0;
        }
        if (!left) {
            int rc = potrf_rec(w, Akk, lda, off0 + k0, cp);
            if (rc) return rc;
            if (below > 0) rc = trsm_rec(below, w, Akk, lda, Akk + w, lda, off0 + k0, cp);
            return rc;
        }
        for (int c0 = 0; c0 < w; c0 += LEAF) {
            const int nj = std::min((int)LEAF, w - c0), mrows = rows - c0;
            double *Pj = Akk + c0 + (size_t)c0 * lda;          // P[c0:, c0:c0+nj]
            int rc;
            if (!right_in && c0 > 0 &&
                (rc = gemm_nt(mrows, nj, c0, -1.0, Akk + c0, lda, Akk + c0, lda, 1.0, Pj, lda, 0, 0, sp)))
                return rc;
            double *invj = cp.inv + (size_t)((off0 + k0 + c0) / LEAF) * LEAF * LEAF;
            hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(LT), 0, sp, nj, Pj, lda, invj, cp.dinfo, off0 + k0 + c0,
                               (int)LEAF_FACTOR, (unsigned long long *)nullptr);
            SGPR_CHECK_LAUNCH();
            if (mrows > nj && (rc = gemm_nt(mrows - nj, nj, nj, 1.0, Pj + nj, lda, invj, LEAF, 0.0, Pj + nj, lda, 0, 0, sp)))
                return rc;
            const int ncols = w - c0 - nj;
            if (right_in && ncols > 0 &&
                (rc = gemm_nt(mrows - nj, ncols, nj, -1.0, Pj + nj, lda, Pj + nj, lda, 1.0, Pj + nj + (size_t)nj * lda, lda, 1, 0, sp)))
                return rc;
        }
        return 0;
    };
    int rc;
    SGPR_HIP(hipEventRecord(ev[2 * nblk], su));                // the side stream joins the caller's stream ...
    SGPR_HIP(hipStreamWaitEvent(sp, ev[2 * nblk], 0));
    StreamJoin join{sp, su, ev[2 * nblk + 1]};                 // ... and leaves it again on every exit path
    gemm_set_overlap(1);                                       // launches below share the device (profile records)
    // Schedule (P = high-priority side stream, U = the caller's stream):
    //   P:  panel(0);  for k:  [after U2(k-1)]  U1(k) = update of block column k+1 by panel k;  panel(k+1)
    //   U:             for k:  [after panel(k)]  U2(k) = update of everything right of block column k+1
    // U1 is a short launch (m x nb x nb) that fills half the chip at best; on the P stream it runs beside
    // the start of U2(k) instead of in front of it, so the U stream does big updates back to back.
    if ((rc = panel(0))) return rc;
    SGPR_HIP(hipEventRecord(ev[0], sp));
    for (int k = 0; k < nblk; ++k) {
        const int k0 = starts[k], w = starts[k + 1] - k0;
        const int k1 = k0 + w;                                 // first row / column of block column k+1
        if (k1 >= n) break;
        const int w1 = starts[k + 2] - k1, k2 = k1 + w1;
        const double *Lk = A + (size_t)k0 * lda;               // panel k: columns k0..k0+w
        // U1 on P: block column k+1 (rows k1.., columns k1..k2); U2(k-1) has touched it before
        if (k > 0) SGPR_HIP(hipStreamWaitEvent(sp, ev[2 * (k - 1) + 1], 0));
        if ((rc = gemm_nt(n - k1, w1, w, -1.0, Lk + k1, lda, Lk + k1, lda, 1.0, A + k1 + (size_t)k1 * lda, lda, 1, 0, sp)))
            return rc;
        // Once the steps are bound by the chain panel -> U1 -> panel and no longer by U2, U2(k) starts only
        // when U1(k) has run: beside a freshly started U2 the short U1 waits for CUs and takes 2-3x longer.
        const double u2_us = (double)(n - k2) * (n - k2) * w / 55e6;
        const bool chain_bound = u2_us < 1.5 * (125.0 * (w1 / LEAF) + 100.0);
        if (chain_bound) SGPR_HIP(hipEventRecord(ev[2 * nblk + 2 + k], sp));
        if ((rc = panel(k + 1))) return rc;
        SGPR_HIP(hipEventRecord(ev[2 * (k + 1)], sp));
        // U2 on U: the rest of the trailing matrix (rows / columns k2..), as soon as panel k is there
        SGPR_HIP(hipStreamWaitEvent(su, chain_bound ? ev[2 * nblk + 2 + k] : ev[2 * k], 0));
        if (k2 < n &&
            (rc = gemm_nt(n - k2, n - k2, w, -1.0, Lk + k2, lda, Lk + k2, lda, 1.0, A + k2 + (size_t)k2 * lda, lda, 1, 0, su)))
            return rc;
        SGPR_HIP(hipEventRecord(ev[2 * k + 1], su));
    }
    return 0;
}

}  // namespace

size_t potrf_batch_flag_bytes(int nbatch) { return ((size_t)nbatch * PFLAG_STRIDE + 1) * sizeof(int); }
int potrf_batch_max_order() { return PW_MAX * (int)LEAF; }

// nbatch lower Cholesky factorisations of order npad (a multiple of 128, <= potrf_batch_max_order()) in one launch;
// `flags`: potrf_batch_flag_bytes(nbatch) bytes of scratch; info[p]: 0 or the 1-based failing minor of problem p
int potrf_batch(int nbatch, int npad, double *A, size_t stride_a, size_t lda, double *inv, size_t stride_inv, int *flags,
                int *info, hipStream_t st)
{
    if (nbatch < 0 || npad <= 0 || npad % LEAF != 0 || npad > PW_MAX * (int)LEAF || lda < (size_t)npad || (lda & 1) != 0 ||
        ((uintptr_t)A & 15) != 0 || (stride_a & 1) != 0) {
        set_error("potrf_batch: bad order / leading dimension / alignment");
        return SGPR_E_ARG;
    }
    if (nbatch == 0) return 0;
    SGPR_HIP(hipMemsetAsync(info, 0, (size_t)nbatch * sizeof(int), st));
    return potrf_batch_panel(nbatch, npad, 0, npad / (int)LEAF, A, stride_a, lda, inv, stride_inv, flags, info, st);
}

// One panel of every problem: columns [k0, k0 + 128 Wd) -- its diagonal block is factored (leaf chain), the rows below it,
// down to row npad, are solved against it.  What a blocked factorisation of the batch composes (batch.hip: two panels with a
// batched rank-k update between them above order 1024).  `flags`: potrf_batch_flag_bytes(nbatch) bytes of scratch of this
// call's own; info is NOT cleared here.
int potrf_batch_panel(int nbatch, int npad, int k0, int Wd, double *A, size_t stride_a, size_t lda, double *inv, size_t stride_inv,
                      int *flags, int *info, hipStream_t st)
{
    const int R = (npad - k0) / (int)LEAF;
    if (nbatch <= 0 || k0 < 0 || k0 % LEAF != 0 || Wd < 1 || Wd > PW_MAX || Wd > R || R > PANEL_G_MAX) {
        set_error("potrf_batch_panel: bad panel");
        return SGPR_E_ARG;
    }
    SGPR_HIP(hipMemsetAsync(flags, 0, potrf_batch_flag_bytes(nbatch), st));
    // workgroups for the strips below the diagonal block: tunable "batch_below" (sgpr_probe_tune), default one per strip
    const int below_max = (int)tune("batch_below", 0);
    const int below = R - Wd <= 0 ? 0 : (below_max > 0 && below_max < R - Wd ? below_max : R - Wd);
    PanelBatch q{A, stride_a, lda, inv, stride_inv, flags, info, flags + (size_t)nbatch * PFLAG_STRIDE, Wd, nbatch, R, k0, Wd + below};
    hipLaunchKernelGGL(panel_batch_kernel, dim3((unsigned)(nbatch * (Wd + below))), dim3(LT), 0, st, q);
    SGPR_CHECK_LAUNCH();
    return 0;
}

// A queue factorisation on `st`'s device has given up (POTRF_HANDOFF_TIMEOUT read back by the caller): no further one is
// started there until the streams are released.  Returns true when the queue was in use (a retry will take the other driver).
bool potrf_queue_mark_failed(hipStream_t st)
{
    // Did THIS thread's last factorisation go through the queue?  (Not "was the queue still on for the device": a give-up of
    // the look-ahead driver's panel kernel at an order the queue never takes must not trigger a pointless second run, and two
    // handles that give up in the same moment must both get their retry, whoever marks the device first.)
    const bool used = t_last_potrf_used_queue;
    if (!used) return false;
    int dev = -1;
    if ((st ? hipStreamGetDevice(st, &dev) : hipGetDevice(&dev)) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return false; }
    QueueDevice &qd = g_qdev[dev];
    std::lock_guard<std::mutex> lock(qd.mu);
    qd.failed = true;
    return true;
}

// the calling thread's pooled events (every device); call with no factorisation of this thread in flight
int potrf_trim()
{
    for (int dev = 0; dev < 64; ++dev) {
        EventPool &p = t_event_pool[dev];
        if (p.ev.empty() || p.top != 0) continue;
        for (hipEvent_t e : p.ev) (void)hipEventDestroy(e);
        p.ev.clear();
    }
    (void)hipGetLastError();
    return 0;
}

int release_device_streams(int dev)
{
    if (dev < 0 || dev >= 64) { set_error("release_device_streams: device index out of range"); return SGPR_E_ARG; }
    return release_streams(dev);
}

int potrf(int n, double *A, size_t lda, void *work, size_t lwork, int *dinfo, hipStream_t st)
{
    if (n < 0 || (n > 0 && lda < (size_t)n)) { set_error("potrf: bad n / lda"); return SGPR_E_ARG; }
    if (lwork < potrf_workspace(n)) { set_error("potrf: workspace too small"); return SGPR_E_ARG; }
    SGPR_HIP(hipMemsetAsync(dinfo, 0, sizeof(int), st));
    if (n == 0) return 0;
    // hand-off flags of the panel kernel (behind the leaf inverses): zero before every factorisation
    int *flags = reinterpret_cast<int *>(static_cast<char *>(work) + inv_bytes(n));
    SGPR_HIP(hipMemsetAsync(flags, 0, flag_bytes(n), st));
    // blocked + look-ahead for mid sizes and for the mid-size blocks of a large recursive
    // factorisation, recursive above (measured crossover; SGPR_POTRF=rec|la overrides)
    static const int mode = [] { const char *e = getenv("SGPR_POTRF"); return !e ? 0 : (e[0] == 'r' ? 1 : 2); }();
    static const int nb_env = [] { const char *e = getenv("SGPR_POTRF_NB"); return e ? atoi(e) : 0; }();
    // blocks of order <= la_max go to the blocked look-ahead driver, larger ones split recursively
    // (SGPR_LA_MAX overrides; measured crossover of round 1)
    static const int la_max_env = [] { const char *e = getenv("SGPR_LA_MAX"); return e ? atoi(e) : 57344; }();
    t_last_potrf_used_queue = false;
    Ctx c{static_cast<double *>(work), dinfo, st, mode == 1 ? 0 : la_max_env, flags};
    if (cholq::ws_bytes(n) > 0) c.qws = static_cast<char *>(work) + inv_bytes(n) + flag_bytes(n) + pub_bytes(n);
    const bool dbg = PANEL_DBG && getenv("SGPR_PANEL_DBG") != nullptr;
    const int T = (n + LEAF - 1) / LEAF;
    if (dbg) {
        (void)hipMalloc((void **)&g_panel_dbg, sizeof(unsigned long long) * 16 * T);
        (void)hipMemset(g_panel_dbg, 0, sizeof(unsigned long long) * 16 * T);
    }
    int rc;
    if (mode == 2 || (nb_env > 0 && mode == 0 && n > 4 * LEAF && n <= c.la_max))
        rc = potrf_lookahead(n, A, lda, c, nb_env > 0 ? nb_env : 0, 0);
    else
        rc = potrf_rec(n, A, lda, 0, c);
    if (dbg) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(16 * (size_t)T);
        (void)hipMemcpy(h.data(), g_panel_dbg, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(g_panel_dbg);
        g_panel_dbg = nullptr;
        std::vector<double> d[9];
        for (int t = 1; t + 1 < T; ++t) {
            const unsigned long long *p = &h[16 * t], *q = &h[16 * (t + 1)];
            if (!p[0] || !p[5] || !q[0]) continue;               // first column of a panel: no chain stamps
            d[0].push_back((p[1] - p[0]) * 0.01);                 // E seen -> solve done
            d[1].push_back((p[2] - p[1]) * 0.01);                 // -> F published (at t == 1 entry)
            d[2].push_back((p[3] - p[2]) * 0.01);                 // -> update done
            d[3].push_back((p[4] - p[3]) * 0.01);                 // -> leaf entered
            d[4].push_back(((double)q[0] - (double)p[4]) * 0.01); // leaf entered -> next strip saw E
            d[5].push_back((p[5] - p[4]) * 0.01);                 // whole leaf incl. inverse + publish
            d[3].back() = ((double)p[4] - (double)p[1]) * 0.01;   // solve stored -> leaf entered (the update)
            d[1].back() = (p[6] - p[0]) * 0.01;                   // E seen -> L staged
            d[2].back() = (p[7] - p[6]) * 0.01;                   // -> MFMA part done
            d[6].push_back((p[11] - p[4]) * 0.01);                // leaf entered -> factor done
            d[7].push_back((p[12] - p[11]) * 0.01);               // -> early flag set
            d[8].push_back(((double)q[2] - (double)p[4]) * 0.01); // leaf entered -> next strip at its wait for E
            if (getenv("SGPR_PANEL_DBG_ALL"))
                fprintf(stderr, "  col %3d: leaf entry -> factor %.1f -> flag %.1f | next strip at wait %.1f, saw E %.1f | its solve %.1f, update %.1f\n", t,
                        (p[11] - p[4]) * 0.01, (p[12] - p[4]) * 0.01, ((double)q[2] - (double)p[4]) * 0.01, ((double)q[0] - (double)p[4]) * 0.01,
                        (q[1] - q[0]) * 0.01, ((double)q[4] - (double)q[1]) * 0.01);
        }
        auto med = [](std::vector<double> &x) { std::sort(x.begin(), x.end()); return x.empty() ? 0.0 : x[x.size() / 2]; };
        fprintf(stderr, "panel chain n=%d (%zu columns): E seen -> solve stored %.1f | E seen -> staged %.1f | -> MFMA part done %.1f | (solve stored -> leaf entry %.1f) | "
                "leaf entry -> next E seen %.1f (factor %.1f, -> flag set %.1f; next strip at its wait %.1f) | whole leaf %.1f us\n", n, d[0].size(), med(d[0]), med(d[1]), med(d[2]), med(d[3]), med(d[4]),
                med(d[6]), med(d[7]), med(d[8]), med(d[5]));
    }
    return rc;
}

int trsm_rlt(int m, int n, const double *L, size_t ldl, double *B, size_t ldb, const void *work,
             hipStream_t st)
{
    if (m <= 0 || n <= 0) return 0;
    Ctx c{const_cast<double *>(static_cast<const double *>(work)), nullptr, st};
    return trsm_rec(m, n, L, ldl, B, ldb, 0, c);
}

// The strip kernels (trsv.hip) take over above one block of the single-workgroup kernel; their ticket /
// progress words live in the flag area behind the leaf inverses (idle once the factor exists).
// SGPR_TRSV=rec keeps the recursive GEMV form (A/B runs).
static bool use_strips(int n, const double *L, size_t ldl)
{
    static const bool off = [] { const char *e = getenv("SGPR_TRSV"); return e && e[0] == 'r'; }();
    return !off && n > TRSV_BLOCK && trsv_strips_ok(n, L, ldl);
}
static int *solve_state(int n, const void *work)
{
    return reinterpret_cast<int *>(const_cast<char *>(static_cast<const char *>(work)) + inv_bytes(n));
}
static double *solve_pub(int n, const void *work)
{
    return reinterpret_cast<double *>(const_cast<char *>(static_cast<const char *>(work)) + inv_bytes(n) + flag_bytes(n));
}

// After a potrs_vec / trsv on this workspace has been waited for: did a strip kernel give up on a hand-off
// (a bounded spin ran out: a bug or a device problem, never a property of the matrix)?  `host8` = the 8 state
// words copied back by the caller (null when the strip kernels were not used for this order).
bool trsv_uses_strips(int n, const double *L, size_t ldl) { return use_strips(n, L, ldl); }

// Waits for the stream and reports a strip solve that gave up on a hand-off (never a property of the matrix).
int solve_status(int n, const double *L, size_t ldl, const void *work, hipStream_t st)
{
    if (n <= 0 || !use_strips(n, L, ldl)) { SGPR_HIP(hipStreamSynchronize(st)); return 0; }
    int h[8] = {};
    SGPR_HIP(hipMemcpyAsync(h, solve_state(n, work), sizeof(h), hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    if (h[2] || h[6]) { set_error("triangular solve: a hand-off between strips timed out"); return SGPR_E_HIP; }
    return 0;
}
const int *trsv_state(int n, const void *work) { return solve_state(n, work); }

int potrs_vec(int n, const double *L, size_t ldl, void *work, double *b, hipStream_t st)
{
    if (n <= 0) return 0;
    Ctx c{const_cast<double *>(static_cast<const double *>(work)), nullptr, st};
    if (use_strips(n, L, ldl)) {
        int *state = solve_state(n, work);
        double *pub = solve_pub(n, work);
        SGPR_HIP(hipMemsetAsync(state, 0, 8 * sizeof(int), st));
        SGPR_HIP(hipMemsetAsync(pub, 0xFF, (size_t)2 * n * sizeof(double), st));
        int rc = trsv_strips(n, L, ldl, c.inv, b, 0, state, pub, st);
        if (rc) return rc;
        return trsv_strips(n, L, ldl, c.inv, b, 1, state + 4, pub + n, st);
    }
    int rc = trsv_n_rec(n, L, ldl, b, 0, c);
    if (rc) return rc;
    return trsv_t_rec(n, L, ldl, b, 0, c);
}

int trsm_rl(int m, int n, const double *L, size_t ldl, double *B, size_t ldb, const void *work, hipStream_t st)
{
    if (m <= 0 || n <= 0) return 0;
    Ctx c{const_cast<double *>(static_cast<const double *>(work)), nullptr, st};
    return trsm_rl_rec(m, n, L, ldl, B, ldb, 0, c);
}

// Blocks of 2 .. 256 right-hand sides take the one-launch strip solves of trsm.hip (L streamed once per 64 columns);
// more than that is compute-bound and stays with the recursion over the grid-wide MFMA kernel.  SGPR_TRSM=rec: always.
bool potrs_mat_uses_strips(int n, int nrhs, const double *L, size_t ldl)
{
    return nrhs >= 2 && nrhs <= 256 && use_strips(n, L, ldl) && trsm_strips_ok(n, L, ldl);
}
size_t potrs_mat_scratch(int n, int nrhs, const double *L, size_t ldl)
{
    if (n <= 0 || nrhs <= 0) return 8;
    return potrs_mat_uses_strips(n, nrhs, L, ldl) ? trsm_strips_scratch(n) : (size_t)n * nrhs * sizeof(double);
}

// B (n x nrhs) := L^-T L^-1 B.  Few right-hand sides: trsm.hip.  Many: through the MFMA kernel, the right-hand sides
// transposed into rows (scratch, nrhs x n), X^T = B^T L^-T (forward) then X^T L^-1 (backward), transposed back.
int potrs_mat(int n, const double *L, size_t ldl, const void *work, double *B, size_t ldb, int nrhs,
              double *scratch, hipStream_t st)
{
    if (n <= 0 || nrhs <= 0) return 0;
    if (potrs_mat_uses_strips(n, nrhs, L, ldl))
        return potrs_strips(n, L, ldl, static_cast<const double *>(work), B, ldb, nrhs, solve_state(n, work), scratch, st);
    int rc = transpose(n, nrhs, B, ldb, scratch, (size_t)nrhs, st);
    if (rc) return rc;
    if ((rc = trsm_rlt(nrhs, n, L, ldl, scratch, (size_t)nrhs, work, st))) return rc;
    if ((rc = trsm_rl(nrhs, n, L, ldl, scratch, (size_t)nrhs, work, st))) return rc;
    return transpose(nrhs, n, scratch, (size_t)nrhs, B, ldb, st);
}

// one-sided solve: b := L^-1 b (trans = 0) or L^-T b (trans = 1)
int trsv(int n, const double *L, size_t ldl, void *work, double *b, int trans, hipStream_t st)
{
    if (n <= 0) return 0;
    Ctx c{const_cast<double *>(static_cast<const double *>(work)), nullptr, st};
    if (use_strips(n, L, ldl)) {
        int *state = solve_state(n, work);
        double *pub = solve_pub(n, work);
        SGPR_HIP(hipMemsetAsync(state, 0, 4 * sizeof(int), st));
        SGPR_HIP(hipMemsetAsync(pub, 0xFF, (size_t)n * sizeof(double), st));
        return trsv_strips(n, L, ldl, c.inv, b, trans, state, pub, st);
    }
    return trans ? trsv_t_rec(n, L, ldl, b, 0, c) : trsv_n_rec(n, L, ldl, b, 0, c);
}

// diagnostic: one leaf factorisation with per-phase cycle counts (load, diag, panel, update,
// write-back, inv diag, inv rows, tail)
int leaf_probe(double *A, size_t lda, double *inv, int *dinfo, unsigned long long *stamps, hipStream_t st)
{
    hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(LT), 0, st, (int)LEAF, A, lda, inv, dinfo, 0, (int)LEAF_FACTOR, stamps);
    SGPR_CHECK_LAUNCH();
    return 0;
}

// leaf inverses of an existing factor (for solves against an L that was not produced by potrf())
int leaf_inverses(int n, const double *L, size_t ldl, void *work, int *dinfo, hipStream_t st)
{
    if (n <= 0) return 0;
    Ctx c{static_cast<double *>(work), dinfo, st};
    return leaves_invert_only(n, const_cast<double *>(L), ldl, c);
}

}  // namespace sgpr

// leaf.h -- the 128 x 128 diagonal leaf of the Cholesky factorisation as a device function: factor and
// invert one block entirely in LDS (one workgroup of 256 threads).  Shared by chol.hip (leaf_kernel and
// the persistent panel kernel) and batch.hip (one small fit per workgroup).
#pragma once
#include "common.h"

namespace sgpr {
namespace leaf {

constexpr int LT = 256;            // threads of the leaf kernel
constexpr int LLD = LEAF + 2;      // LDS leading dimension: even (16-B aligned column pairs), 4*LLD mod 64 banks = 8
constexpr int PW = 16;             // panel width inside the leaf

enum { LEAF_FACTOR = 0, LEAF_INVERT_ONLY = 1 };

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

// agent-scope write-through (sc1) 8-byte store
__device__ __forceinline__ void store_wt(double *p, double v)
{
    __hip_atomic_store((__attribute__((address_space(1))) unsigned long long *)p, (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A (nb x nb, lower, global) -> L in place (mode FACTOR) and inv(L) -> inv (LEAF x LEAF,
// ld LEAF, zero-filled outside the nb x nb lower triangle).
//
// One workgroup, the whole block in LDS (128 x 130 fp64 = 130 KiB), padded to 128 with an
// identity so every loop bound is a compile-time constant.  Both phases work on 16-column
// panels (8 panel steps, 3 barriers each) instead of one barrier-separated step per column:
//   factor : 16x16 diagonal block by one wave (row per lane, pivots/columns via shuffles) ->
//            panel rows solved one per thread against it -> rank-16 update of the trailing
//            lower triangle in 4x4 register tiles;
//   inverse: the eight 16x16 diagonal blocks at once, then recursive doubling (16 -> 32 -> 64 -> 128):
//            X21 = -X22 L21 X11 for every pair of a level on the matrix cores.
__device__ __forceinline__ void leaf_body(double *s /* LEAF_LDS doubles: the block, then two PW x (PW + 1) buffers */, int nb,
                                          double *A, size_t lda, double *inv, int *dinfo, int goff, int mode,
                                          unsigned long long *stamps, int *early_flag = nullptr, bool preloaded = false,
                                          unsigned long long *rt = nullptr /* diagnostic: 100 MHz stamps (factor done, early flag set) */)
{
    double *const sInv = s + LEAF * LLD;   // inv(L11) of the current panel and of the next one (double-buffered)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // Rows / columns from nend on are the identity padding: nothing is computed there (a batched fit of order 80
    // runs 4 of the 7 panel steps).  nb == LEAF is a compile-time constant at the call sites that matter.
    const int nend = (nb + PW - 1) / PW * PW;
    // diagnostic phase clock (stamps == nullptr in production): cycles per phase, summed over panels
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = stamps ? __builtin_amdgcn_s_memtime() : 0;
    auto mark = [&](int i) {
        if (stamps) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            ph[i] += t - tprev;
            tprev = t;
        }
    };

    if (preloaded) {
        // the caller has put the tile into s (strict upper triangle zero) behind a barrier: chol.hip, the chain
    } else if (nb == LEAF && (lda & 1) == 0 && (((uintptr_t)A & 15) == 0)) {
        // full leaf: 32 independent 16-B loads per thread (the whole square is read, the strict
        // upper triangle -- whatever it holds -- is replaced by zeros on the way into LDS)
#pragma unroll 8
        for (int it = 0; it < LEAF * LEAF / (2 * LT); ++it) {
            const int idx = it * LT + tid;
            const int i = 2 * (idx % (LEAF / 2)), c = idx / (LEAF / 2);
            const double2_t v = *reinterpret_cast<const double2_t *>(A + (size_t)i + (size_t)c * lda);
            s[c * LLD + i] = (i >= c) ? v.x : 0.0;
            s[c * LLD + i + 1] = (i + 1 >= c) ? v.y : 0.0;
        }
    } else {
        for (int idx = tid; idx < LEAF * LEAF; idx += LT) {
            const int i = idx % LEAF, c = idx / LEAF;
            double v = (i == c) ? 1.0 : 0.0;               // identity padding beyond nb
            if (i < nb && c < nb) v = (i >= c) ? A[(size_t)i + (size_t)c * lda] : 0.0;
            s[c * LLD + i] = v;
        }
    }
    __syncthreads();
    mark(0);

    if (mode == LEAF_FACTOR) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, l4 = lane >> 4;
        // L goes out to A as its entries become final (write-through when it is handed over early)
        auto put = [&](double *dst, double v) {
            if (early_flag) store_wt(dst, v);
            else *dst = v;
        };
        auto bcast = [&](double v, int ln) {              // v of lane ln, wave-uniform (compile-time ln)
            const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(v), ln);
            const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(v), ln);
            return __hiloint2double((int)hi, (int)lo);
        };
        double mneg[PW];                                  // A-operand mask of column step j of (A), as a factor
#pragma unroll
        for (int j = 0; j < PW; ++j) mneg[j] = (l4 == (j & 3) && l15 > j) ? -1.0 : 0.0;
        auto rsqrt_nr = [&](double d) {                   // hardware estimate + two Newton steps
            double rl = __builtin_amdgcn_rsq(d);
            rl = rl * __builtin_fma(-0.5 * d * rl, rl, 1.5);
            rl = rl * __builtin_fma(-0.5 * d * rl, rl, 1.5);
            return rl;
        };
        // The 16 x 16 diagonal block at c0 as a FULL symmetric tile in the accumulator layout of v_mfma_f64_16x16x4
        // (element r of lane (l15, l4) = D(4 r + l4, l15)), mirrored out of the lower triangle of s.
        auto load_diag = [&](int c0) {
            double4_t D;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * r + l4, j = l15;
                D[r] = s[(c0 + (i >= j ? j : i)) * LLD + c0 + (i >= j ? i : j)];
            }
            return D;
        };
        // (A) the diagonal block, ONE wave, in registers from start to end.  Column j of a symmetric tile is its row j, and
        // row j sits in the lanes l4 == (j & 3), element j >> 2, at l15 = row index: exactly where BOTH operands of a
        // rank-1 v_mfma_f64_16x16x4 (k-slot j & 3) take it from.  So a column step is: scale by 1/sqrt(pivot), one MFMA --
        // no cross-lane traffic, no LDS round trip; the next pivot (two v_readlane pairs + its 1/sqrt chain) runs beside
        // the MFMA.  Only the A operand is masked (k-slot and rows below j: mneg[j] = -1 there, 0 elsewhere): whatever
        // the other lanes hold is multiplied by zero, and rows / columns <= j of D are dead.  An identity tile takes the
        // same eliminations with its rows scaled at the end: it ends as inv(L11), which is what the panel rows are
        // multiplied by (B) and what the early hand-off / the inverse phase start from.
        auto diag_factor = [&](int c0, double4_t D, double *sI) {
            double4_t X;
#pragma unroll
            for (int r = 0; r < 4; ++r) X[r] = (4 * r + l4 == l15) ? 1.0 : 0.0;
            double rl = rsqrt_nr(bcast(D[0], 0));
            double lv[PW], rls[PW];                                            // kept in registers: LDS traffic inside the loop
#pragma unroll                                                                 // cost 100 cycles per column (tools/probe_lat.py)
            for (int j = 0; j < PW; ++j) {
                const int g = j & 3, rj = j >> 2;
                const double v = D[rj] * rl;                                   // lanes l4 == g: L(c0 + l15, c0 + j)
                const double x = X[rj] * rl;                                   // lanes l4 == g: row j of inv(L11)
                double rn = 0.0;
                if (j + 1 < PW) {
                    const double dnext = bcast(D[(j + 1) >> 2], (j + 1) + 16 * ((j + 1) & 3));
                    const double lnext = bcast(v, (j + 1) + 16 * g);
                    rn = rsqrt_nr(__builtin_fma(-lnext, lnext, dnext));
                }
                lv[j] = v;
                rls[j] = rl;
                const double nv = v * mneg[j];
                D = __builtin_amdgcn_mfma_f64_16x16x4f64(nv, v, D, 0, 0, 0);
                X = __builtin_amdgcn_mfma_f64_16x16x4f64(nv, x, X, 0, 0, 0);
                rl = rn;
            }
            // L11 out to LDS: column j sits in the lanes l4 == (j & 3) (above the diagonal: dead values, never read)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (l4 == g) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) s[(c0 + 4 * m + g) * LLD + c0 + l15] = lv[4 * m + g];
                }
            }
            // inv(L11): row c of the identity tile times 1 / L11(c, c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double rr = l4 == 0 ? rls[4 * r] : l4 == 1 ? rls[4 * r + 1] : l4 == 2 ? rls[4 * r + 2] : rls[4 * r + 3];
                sI[(4 * r + l4) * (PW + 1) + l15] = X[r] * rr;                 // sI[c][i] = inv(L11)(c, i)
            }
            // a pivot that was not positive leaves NaN from its column on: LAPACK's info is the first such column
            const double ljj = s[(c0 + l15) * LLD + c0 + l15];
            const unsigned long long okm = __ballot(ljj > 0.0) & 0xffffull;
            if (okm != 0xffffull && lane == 0 && *dinfo == 0) *dinfo = goff + c0 + __builtin_ctzll(~okm) + 1;
        };
        // (B) sixteen panel rows at row0: X = R inv(L11)^T, four MFMAs
        auto solve_tile = [&](int c0, int row0, const double *sI) {
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < PW / 4; ++kk)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s[(c0 + 4 * kk + l4) * LLD + row0 + l15], sI[l15 * (PW + 1) + 4 * kk + l4],
                                                           acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[(c0 + l15) * LLD + row0 + 4 * r + l4] = acc[r];
        };
        // (C) one 16x16 tile of the trailing update on the matrix cores: k = 16 = four
        // v_mfma_f64_16x16x4_f64; both operands are "row contiguous, k strided" reads of the
        // panel columns (A: lane&15 = i, B: lane&15 = j, lane>>4 = k).
        auto update_tile = [&](int c0, int i0, int j0) {
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < PW / 4; ++kk) {
                const double *col = s + (c0 + 4 * kk + l4) * LLD;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(col[i0 + l15], col[j0 + l15], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s[(j0 + l15) * LLD + i0 + 4 * r + l4] -= acc[r];  // D(i,j): j = lane&15, i = 4r + lane>>4
        };
        // The finished columns c0 .. c0+15 go out to A (threads t of nth; rows fastest: 512-byte runs), and the diagonal
        // block's place in s takes inv(L11): after the last panel s is what the inverse phase starts from.
        auto put_panel = [&](int c0, int t, int nth, const double *sI) {
            for (int e = t; e < PW * PW; e += nth) {
                const int i = e & 15, c = e >> 4;
                if (i >= c) {
                    if (c0 + i < nb) put(A + (size_t)(c0 + i) + (size_t)(c0 + c) * lda, s[(c0 + c) * LLD + c0 + i]);
                    s[(c0 + c) * LLD + c0 + i] = sI[i * (PW + 1) + c];
                }
            }
            const int top = min(nend, nb);
            for (int c = t >> 6; c < PW; c += nth >> 6)           // (nth is a multiple of 64: a wave per column, 64 rows a go)
                for (int i = c0 + PW + (t & 63); i < top; i += 64) put(A + (size_t)i + (size_t)(c0 + c) * lda, s[(c0 + c) * LLD + i]);
        };

        if (wave == 0) diag_factor(0, load_diag(0), sInv);
        __syncthreads();
        mark(1);
        for (int c0 = 0; c0 < nend - PW; c0 += PW) {
            const int r0 = c0 + PW;
            const int rem = nend - r0;
            const int nt = rem / 16;
            const double *sI = sInv + ((c0 / PW) & 1) * PW * (PW + 1);
            double *sIn = sInv + (((c0 / PW) & 1) ^ 1) * PW * (PW + 1);
            // ---- (B) panel rows on the matrix cores.  Wave 0 takes the rows of the NEXT diagonal block and goes on to update
            // that block in its registers (its own X out of LDS again as both operands): it enters (A) of the next panel
            // right behind the barrier.
            double4_t Dn = {0.0, 0.0, 0.0, 0.0};
            if (wave == 0) {
                solve_tile(c0, r0, sI);
                Dn = load_diag(r0);
#pragma unroll
                for (int kk = 0; kk < PW / 4; ++kk) {
                    const double xa = s[(c0 + 4 * kk + l4) * LLD + r0 + l15];
                    Dn = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa, xa, Dn, 0, 0, 0);
                }
            } else {
                for (int t = wave; t < nt; t += LT / 64 - 1) solve_tile(c0, r0 + 16 * t, sI);
            }
            __syncthreads();
            mark(2);
            // ---- (C) trailing update by waves 1-3, with the NEXT diagonal block factored underneath it by wave 0
            if (wave == 0) {
                diag_factor(r0, Dn, sIn);
            } else {
                put_panel(c0, tid - 64, LT - 64, sI);
                // lower tiles (ti, tj) in row-major order, every third one from this wave's start; (0, 0) is wave 0's
                int ti = 1, tj = wave - 1;                        // positions 1, 2, 3 of the order are (1,0), (1,1), (2,0)
                if (tj > ti) { tj = 0; ++ti; }
                while (ti < nt) {
                    update_tile(c0, r0 + 16 * ti, r0 + 16 * tj);
                    tj += LT / 64 - 1;
                    while (tj > ti) { tj -= ti + 1; ++ti; }
                }
            }
            __syncthreads();
            mark(3);
        }
        put_panel(nend - PW, tid, LT, sInv + (((nend - PW) / PW) & 1) * PW * (PW + 1));
        __syncthreads();
        mark(4);
        if (rt && tid == 0) rt[0] = __builtin_amdgcn_s_memrealtime();
    }

    // ---- inverse (LAPACK dtrtri order, last panel first); only the lower triangle of s is read.
    // (I0) all eight 16x16 diagonal blocks are inverted at once, in place: 128 threads, one column
    //      of one block each (x = solve L11 x = e_c), values held in registers across the barrier.
    if (mode != LEAF_FACTOR) {        // (the factorisation leaves inv(L11) in place of every diagonal block by itself)
        const int blk = tid >> 4, c = tid & 15, d0 = blk * PW;
        double x[PW];
        if (tid < 128) {
            // right-looking substitution: x_k is final after one multiply by 1 / L_kk (all sixteen reciprocals are
            // independent and go first), and its updates of the later entries are independent of each other
            double rd[PW];
#pragma unroll
            for (int k = 0; k < PW; ++k) rd[k] = 1.0 / s[(d0 + k) * LLD + d0 + k];
#pragma unroll
            for (int i = 0; i < PW; ++i) x[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < PW; ++k) {
                x[k] *= rd[k];
#pragma unroll
                for (int i = k + 1; i < PW; ++i) x[i] = __builtin_fma(-s[(d0 + k) * LLD + d0 + i], x[k], x[i]);
            }
        }
        __syncthreads();
        if (tid < 128) {
#pragma unroll
            for (int i = 0; i < PW; ++i)
                if (i >= c) s[(d0 + c) * LLD + d0 + i] = x[i];
        }
        __syncthreads();
    }
    if (early_flag) {
        // L (in A) and the inverses of its eight 16 x 16 diagonal blocks -- which ARE the diagonal blocks of
        // inv(L) -- are all a blocked triangular solve against L needs (chol.hip, panel_trsm): hand them over
        // now, ~20 us before the full inverse (agent-scope release as in panel_publish)
        for (int idx = tid; idx < LEAF * PW; idx += LT) {
            const int i = idx % LEAF, c = (i / PW) * PW + idx / LEAF;     // row i, the 16 columns of its diagonal block
            store_wt(inv + (size_t)i + (size_t)c * LEAF, (i >= c) ? s[c * LLD + i] : 0.0);
        }
        // every byte handed over here (L above, these blocks) went out write-through: drained stores, the
        // workgroup's barrier, then the flag -- no release fence (MI355X_MICROARCH.md, publish-large: 3 vs 8 us)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store((__attribute__((address_space(1))) int *)early_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (rt && tid == 0) rt[1] = __builtin_amdgcn_s_memrealtime();
    }
    mark(5);
    // (I1) recursive doubling: with the diagonal blocks of size b inverted, the blocks of size 2b follow
    //      from X21 = -X22 L21 X11 for every pair at once -- three levels (b = 16, 32, 64), two small
    //      matrix products each, instead of seven dependent panel steps (12.4 -> ~6 us).  The product
    //      T = X22 L21 is parked in the strictly upper corner of s (rows 0..63, columns 64..127),
    //      which nothing reads: the write-back below masks the upper triangle.
    {
        const int wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
        double *const scr = s + (LEAF / 2) * LLD;                 // scr[c * LLD + r], r < 64, c < 64
        for (int b = PW; b < LEAF; b *= 2) {
            const int tb = b / 16, per = tb * tb, ntiles = (LEAF / (2 * b)) * per;
            // T = X22 L21 (X22 lower triangular: k <= row)
            for (int t = wave; t < ntiles; t += LT / 64) {
                const int p = t / per, tt = t - p * per, ti = tb - 1 - tt % tb, tj = tt / tb;   // heavy row tiles first
                const int o = 2 * b * p, o2 = o + b;
                if (o2 + 16 * ti >= nend) continue;   // rows of the padding: L21 is zero there, and so stays X21
                const int row = 16 * ti + l15;
                double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                for (int kb = 0; kb < 16 * (ti + 1); kb += 16) {
                    double xa[4], lb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = kb + 4 * u + l4;
                        xa[u] = (k <= row) ? s[(o2 + k) * LLD + o2 + row] : 0.0;
                        lb[u] = s[(o + 16 * tj + l15) * LLD + o2 + k];
                    }
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[0], lb[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[1], lb[1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2], lb[2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[3], lb[3], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[(16 * tj + l15) * LLD + b * p + 16 * ti + 4 * r + l4] = acc0[r] + acc1[r];
            }
            __syncthreads();
            mark(6);
            // X21 = -T X11 (X11 lower triangular: k >= column), written over L21
            for (int t = wave; t < ntiles; t += LT / 64) {
                const int p = t / per, tt = t - p * per, ti = tt % tb, tj = tt / tb;
                const int o = 2 * b * p, o2 = o + b;
                if (o2 + 16 * ti >= nend) continue;
                const int col = 16 * tj + l15;
                double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                for (int kb = 16 * tj; kb < b; kb += 16) {
                    double ta[4], xb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = kb + 4 * u + l4;
                        ta[u] = scr[k * LLD + b * p + 16 * ti + l15];
                        xb[u] = (k >= col) ? s[(o + col) * LLD + o + k] : 0.0;
                    }
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[0], xb[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[1], xb[1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[2], xb[2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[3], xb[3], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) s[(o + col) * LLD + o2 + 16 * ti + 4 * r + l4] = -(acc0[r] + acc1[r]);
            }
            __syncthreads();
            mark(5);
        }
    }
#pragma unroll 8
    for (int it = 0; it < LEAF * LEAF / (2 * LT); ++it) {
        const int idx = it * LT + tid;
        const int i = 2 * (idx % (LEAF / 2)), c = idx / (LEAF / 2);
        const double v0 = (i >= c && i < nb && c < nb) ? s[c * LLD + i] : 0.0;
        const double v1 = (i + 1 >= c && i + 1 < nb && c < nb) ? s[c * LLD + i + 1] : 0.0;
        *reinterpret_cast<double2_t *>(inv + (size_t)i + (size_t)c * LEAF) = double2_t{v0, v1};
    }
    mark(7);
    if (stamps && tid == 0)
        for (int i = 0; i < 8; ++i) stamps[i] = ph[i];
}

constexpr int LEAF_LDS = LEAF * LLD + 2 * PW * (PW + 1);   // doubles of LDS leaf_body needs

}  // namespace leaf
}  // namespace sgpr

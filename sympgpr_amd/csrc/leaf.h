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
__device__ __forceinline__ void leaf_body(double *s /* LEAF * LLD */, double *sInv /* PW * (PW + 1) */,
                                          double *sRl /* PW: 1 / L11(j,j) of the current panel */, int nb,
                                          double *A, size_t lda, double *inv, int *dinfo, int goff, int mode,
                                          unsigned long long *stamps, int *early_flag = nullptr, bool preloaded = false)
{
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
        const int wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
        // (A) 16x16 diagonal block at c0, one wave: lane r holds row r in registers; the pivot of
        // column j comes from lane j by v_readlane, 1/sqrt by the hardware estimate + two Newton
        // steps, the scaled column is published through a 16-double LDS strip and read back as
        // broadcasts (15 independent reads instead of a chain of dependent cross-lane shuffles).
        // L goes out to A as its entries become final (write-through when it is handed over early)
        auto put = [&](double *dst, double v) {
            if (early_flag) store_wt(dst, v);
            else *dst = v;
        };
        // the factored 16 x 16 diagonal block at c0, two entries per helper thread t < 128, out of LDS
        auto put_diag = [&](int c0, int t) {
            const int i = t & 15, c = (t >> 4) * 2;
            if (c0 + i < nb) {
                if (c <= i) put(A + (size_t)(c0 + i) + (size_t)(c0 + c) * lda, s[(c0 + c) * LLD + c0 + i]);
                if (c + 1 <= i) put(A + (size_t)(c0 + i) + (size_t)(c0 + c + 1) * lda, s[(c0 + c + 1) * LLD + c0 + i]);
            }
        };
        auto diag_factor = [&](int c0) {
            double a[PW];
#pragma unroll
            for (int c = 0; c < PW; ++c)
                a[c] = (lane < PW && c <= lane) ? s[(c0 + c) * LLD + c0 + lane] : 0.0;
            bool bad = false;
            int badj = 0;
            auto pivot = [&](double v, int j, double &d, double &rl) {   // d = v on lane j; rl = 1/sqrt(d)
                const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(v), j);
                const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(v), j);
                d = __hiloint2double((int)hi, (int)lo);
                rl = __builtin_amdgcn_rsq(d);
                rl = rl * __builtin_fma(-0.5 * d * rl, rl, 1.5);
                rl = rl * __builtin_fma(-0.5 * d * rl, rl, 1.5);
            };
            double d, rl;
            pivot(a[0], 0, d, rl);
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                if (!(d > 0.0) && !bad) { bad = true; badj = j; }
                a[j] = (lane == j) ? d * rl : a[j] * rl;
                if (lane == j) sRl[j] = rl;
                // the next pivot needs only lane j+1's own values (a[j+1] - a[j]^2): its 1/sqrt chain
                // runs beside the column update below
                double dn = 0.0, rn = 0.0;
                if (j + 1 < PW) pivot(__builtin_fma(-a[j], a[j], a[j + 1]), j + 1, dn, rn);
                // column update a(r,c) -= L(r,j) L(c,j): L(c,j) is a[j] of lane c -- read across the wave into
                // scalar registers (v_readlane), not through an LDS strip and its write -> wait -> read round trip
#pragma unroll
                for (int c = j + 1; c < PW; ++c) {
                    const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(a[j]), c);
                    const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(a[j]), c);
                    a[c] = __builtin_fma(-a[j], __hiloint2double((int)hi, (int)lo), a[c]);
                }
                d = dn;
                rl = rn;
            }
            if (bad && lane == 0 && *dinfo == 0) *dinfo = goff + c0 + badj + 1;
            if (lane < PW) {
#pragma unroll
                for (int c = 0; c < PW; ++c)
                    if (c <= lane) s[(c0 + c) * LLD + c0 + lane] = a[c];
            }
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

        if (wave == 0) diag_factor(0);
        __syncthreads();
        mark(1);
        for (int c0 = 0; c0 < nend - PW; c0 += PW) {
            const int r0 = c0 + PW;
            const int rem = nend - r0;
            // ---- (B) panel rows: r := r L11^-T, one row per thread
            if (tid < rem) {
                const int i = r0 + tid;
                double r[PW];
#pragma unroll
                for (int c = 0; c < PW; ++c) r[c] = s[(c0 + c) * LLD + i];
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    double acc = r[j];
#pragma unroll
                    for (int k = 0; k < j; ++k) acc = __builtin_fma(-r[k], s[(c0 + k) * LLD + c0 + j], acc);
                    r[j] = acc * sRl[j];
                }
#pragma unroll
                for (int c = 0; c < PW; ++c) s[(c0 + c) * LLD + i] = r[c];
                if (i < nb) {
#pragma unroll
                    for (int c = 0; c < PW; ++c) put(A + (size_t)i + (size_t)(c0 + c) * lda, r[c]);
                }
            } else if (tid >= LT / 2) {
                put_diag(c0, tid - LT / 2);            // threads 128.. never have a row here: the diagonal block goes out
            }
            __syncthreads();
            mark(2);
            // ---- (C) trailing update, with the NEXT diagonal block factored underneath it: wave 0
            // updates tile (0,0) first and goes straight on to (A) of the next panel while waves
            // 1-3 update the other tiles.
            const int nt = rem / 16;
            const int ntile = nt * (nt + 1) / 2;
            if (wave == 0) {
                update_tile(c0, r0, r0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                diag_factor(r0);
            } else {
                for (int idx = wave; idx < ntile; idx += LT / 64 - 1) {
                    int ti = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
                    while (ti * (ti + 1) / 2 > idx) --ti;
                    while ((ti + 1) * (ti + 2) / 2 <= idx) ++ti;
                    const int tj = idx - ti * (ti + 1) / 2;
                    update_tile(c0, r0 + 16 * ti, r0 + 16 * tj);
                }
            }
            __syncthreads();
            mark(3);
        }
        if (tid >= LT / 2) put_diag(nend - PW, tid - LT / 2);
        __syncthreads();
        mark(4);
    }

    // ---- inverse (LAPACK dtrtri order, last panel first); only the lower triangle of s is read.
    // (I0) all eight 16x16 diagonal blocks are inverted at once, in place: 128 threads, one column
    //      of one block each (x = solve L11 x = e_c), values held in registers across the barrier.
    {
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

constexpr int LEAF_LDS = LEAF * LLD + PW * (PW + 1) + PW;   // doubles of LDS leaf_body needs

}  // namespace leaf
}  // namespace sgpr

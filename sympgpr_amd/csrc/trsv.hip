// trsv.hip -- one-right-hand-side triangular solves  b := L^-1 b  /  b := L^-T b  as ONE launch each.
//
// Replaces the two scipy.linalg.solve_triangular calls of solve_cholesky
// (python/functions/func.py:174-177 -> LAPACK dtrtrs) for a single right-hand side.
//
// Roofline: HBM read bandwidth -- L is read once per solve (8 n^2 / 2 bytes).  Round 1 recursed like the
// factorisation (GEMV launches + a single-workgroup 512-row block kernel): 1024 dependent launches of
// ~50 us at n = 131072, 2.3 TB/s.  Here every 128-row strip (forward) / 128-column strip (backward) is
// owned by one workgroup that streams its tiles of L at full width while the strips before it are still
// being solved, and picks up their results through a progress counter:
//
//   forward, strip r:   y_r = inv(L_rr) (b_r - sum_{c<r} L(r,c) y_c)      tiles (r, 0..r-1), c ascending
//   backward, strip c:  x_c = inv(L_cc)^T (y_c - sum_{r>c} L(r,c)^T x_r)  tiles (T-1..c+1, c), r descending
//
// Strips are handed out in dependency order by a ticket (a workgroup that waits always waits for one that
// started earlier, whatever the residency), finished strips advance `ready` (strip s done <=> all strips
// before it in ticket order done), and a workgroup only polls when it has caught up with the frontier.
// Hand-off form (MI355X_MICROARCH.md, inter-workgroup visibility, "Valid forms", 8-byte granules): the
// solved segment IS the flag.  It is published into a side buffer `pub` that the host pre-fills with a
// sentinel bit pattern (all ones: a NaN no arithmetic produces; a NaN result is canonicalised before it is
// published), with agent-scope (sc1, write-through) 8-byte stores by ONE wave; the 128 lanes of the consumer
// that carry the segment re-load their own element (sc1) until it is no longer the sentinel -- no flag,
// no fence, no drain between data and signal on the chain of strips.  The progress counter (written behind
// a drain, off that chain) only tells later strips how far they may read without polling.
// L itself and the leaf inverses are read-only here: plain 16-byte loads.
#include "common.h"

namespace sgpr {

namespace {

constexpr int TS_T = 512;                   // threads: 8 waves, wave w owns columns [16 w, 16 w + 16) of a tile
constexpr int TS_W = TS_T / 64;
constexpr int TS_CPW = LEAF / TS_W;         // 16 columns of a tile per wave
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) int gi32;

struct TrsvArgs {
    int T;                 // strips (n = 128 T)
    const double *L;
    size_t ldl;
    const double *inv;     // leaf inverses, LEAF x LEAF each
    double *b;             // right-hand side in, solution out
    double *pub;           // n doubles, all-ones on entry: the solved segments as the strips publish them
    int *state;            // [0] ticket, [1] ready (strips finished, in ticket order), [2] timeout flag
    int trans;
};

__device__ __forceinline__ void store_sc1(double *p, double v)
{
    __hip_atomic_store((gu64 *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_sc1(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
constexpr unsigned long long UNPUBLISHED = ~0ull;
__device__ __forceinline__ void publish(double *p, double v)
{
    if (v != v) v = __longlong_as_double(0x7FF8000000000000ll);       // never the sentinel
    store_sc1(p, v);
}

// one 128 x 128 tile (column-major, leading dimension ld) into registers: lane l of wave w holds rows
// (2l, 2l+1) of the wave's 16 columns
__device__ __forceinline__ void load_tile(const double *tile, size_t ld, int lane, int wave, double2_t (&reg)[TS_CPW])
{
    const double *p = tile + 2 * lane + (size_t)(wave * TS_CPW) * ld;
#pragma unroll
    for (int j = 0; j < TS_CPW; ++j) reg[j] = *reinterpret_cast<const double2_t *>(p + (size_t)j * ld);
}

__global__ __launch_bounds__(TS_T) void trsv_strips_kernel(const TrsvArgs a)
{
    __shared__ double vec[2][LEAF];          // the segment of the solution a tile is multiplied with
    __shared__ double red[TS_W][LEAF];       // cross-wave partial sums (forward)
    __shared__ double red2[TS_W][LEAF];      // second exchange (the diagonal leaf product)
    __shared__ double tv[LEAF];              // b_strip - sum
    __shared__ int sh[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = a.T;
    const bool fwd = a.trans == 0;
    int known = 0;                           // strips known to be finished (ticket order)
    for (;;) {
        if (tid == 0) sh[0] = atomicAdd(a.state, 1);
        __syncthreads();
        const int tk = sh[0];
        __syncthreads();
        if (tk >= T) return;
        const int s = fwd ? tk : T - 1 - tk;                 // this workgroup's strip
        // tile q = 0..tk-1 of the strip, in dependency order: forward (s, q); backward (T-1-q, s).
        // Segment needed by tile q = the strip finished q-th.
        auto tile_ptr = [&](int q) {
            return fwd ? a.L + (size_t)s * LEAF + (size_t)q * LEAF * a.ldl
                       : a.L + (size_t)(T - 1 - q) * LEAF + (size_t)s * LEAF * a.ldl;
        };
        auto seg_ptr = [&](int q) { return a.pub + (size_t)(fwd ? q : T - 1 - q) * LEAF; };
        double acc_r0 = 0.0, acc_r1 = 0.0;                    // forward: sums of this lane's two rows
        double acc_c[TS_CPW];                                 // backward: sums of this wave's 16 columns
#pragma unroll
        for (int j = 0; j < TS_CPW; ++j) acc_c[j] = 0.0;
        double2_t cur[TS_CPW], nxt[TS_CPW];
        double seg_next = 0.0;
        bool have_next = false;
        const double *inv_tile = a.inv + (size_t)s * LEAF * LEAF;
        if (tk > 0) load_tile(tile_ptr(0), a.ldl, lane, wave, cur);
        else load_tile(inv_tile, LEAF, lane, wave, cur);
        for (int q = 0; q < tk; ++q) {
            // the next tile's loads go out before anything that may wait; behind the last tile comes the
            // inverted diagonal leaf, so that its latency is not paid on the chain of strips
            if (q + 1 < tk) load_tile(tile_ptr(q + 1), a.ldl, lane, wave, nxt);
            else load_tile(inv_tile, LEAF, lane, wave, nxt);
            // segment q: fetched one step ahead, or (at the frontier) polled for: the 128 lanes that carry
            // the segment poll the counter themselves and load their element the moment it moves
            if (q >= known) {
                if (tid < LEAF) {
                    unsigned spins = 0;
                    unsigned long long bits;
                    gu64 *src = (gu64 *)(seg_ptr(q) + tid);
                    while ((bits = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == UNPUBLISHED) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (20u << 20)) break;                    // ~ seconds: give up
                    }
                    vec[q & 1][tid] = __longlong_as_double((long long)bits);
                    if (bits == UNPUBLISHED) __hip_atomic_store((gi32 *)(a.state + 2), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (tid == 0)   // how far the others are: later segments up to there need no polling
                        sh[2 + (q & 1)] = bits == UNPUBLISHED ? -1 : __hip_atomic_load((gi32 *)(a.state + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                const int r = sh[2 + (q & 1)];
                if (r < 0) return;
                known = r > q + 1 ? r : q + 1;
            } else {
                const double seg = have_next ? seg_next : (tid < LEAF ? load_sc1(seg_ptr(q) + tid) : 0.0);
                if (tid < LEAF) vec[q & 1][tid] = seg;
                __syncthreads();
            }
            // prefetch the following segment when it is known to be there
            have_next = (q + 1 < tk) && (q + 1 < known);
            if (have_next && tid < LEAF) seg_next = load_sc1(seg_ptr(q + 1) + tid);
            const double *v = vec[q & 1];
            if (fwd) {
#pragma unroll
                for (int j = 0; j < TS_CPW; ++j) {
                    const double y = v[wave * TS_CPW + j];
                    acc_r0 = __builtin_fma(cur[j].x, y, acc_r0);
                    acc_r1 = __builtin_fma(cur[j].y, y, acc_r1);
                }
            } else {
                const double x0 = v[2 * lane], x1 = v[2 * lane + 1];
#pragma unroll
                for (int j = 0; j < TS_CPW; ++j) acc_c[j] = __builtin_fma(cur[j].x, x0, __builtin_fma(cur[j].y, x1, acc_c[j]));
            }
#pragma unroll
            for (int j = 0; j < TS_CPW; ++j) cur[j] = nxt[j];
        }
        // ---- the strip's own segment: t = b_s - sum, then multiply with the inverted diagonal leaf (in cur)
        double *bs = a.b + (size_t)s * LEAF;
        if (fwd) {
            red[wave][2 * lane] = acc_r0;
            red[wave][2 * lane + 1] = acc_r1;
            __syncthreads();
            if (tid < LEAF) {
                double r = 0.0;
#pragma unroll
                for (int w = 0; w < TS_W; ++w) r += red[w][tid];
                tv[tid] = bs[tid] - r;
            }
            __syncthreads();
            // y = inv t: row sums over this wave's 16 columns of inv
            double r0 = 0.0, r1 = 0.0;
#pragma unroll
            for (int j = 0; j < TS_CPW; ++j) {
                const double t = tv[wave * TS_CPW + j];
                r0 = __builtin_fma(cur[j].x, t, r0);
                r1 = __builtin_fma(cur[j].y, t, r1);
            }
            red2[wave][2 * lane] = r0;
            red2[wave][2 * lane + 1] = r1;
            __syncthreads();
            // the final sums and the hand-off are ONE wave's job: store, drain, signal -- no further barrier
            if (wave == 0) {
                double y0 = 0.0, y1 = 0.0;
#pragma unroll
                for (int w = 0; w < TS_W; ++w) { y0 += red2[w][2 * lane]; y1 += red2[w][2 * lane + 1]; }
                double *ps = a.pub + (size_t)s * LEAF;
                publish(ps + 2 * lane, y0);
                publish(ps + 2 * lane + 1, y1);
                bs[2 * lane] = y0;
                bs[2 * lane + 1] = y1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store((gi32 *)(a.state + 1), tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            // column sums: reduce each of the 16 accumulators over the wave's 64 lanes
#pragma unroll
            for (int j = 0; j < TS_CPW; ++j) {
                double v = acc_c[j];
                for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
                if (lane == 0) tv[wave * TS_CPW + j] = bs[wave * TS_CPW + j] - v;
            }
            __syncthreads();
            // x = inv^T t: column sums of inv against t (rows 2l, 2l+1 of this lane)
            const double t0 = tv[2 * lane], t1 = tv[2 * lane + 1];
            double xs[TS_CPW];
#pragma unroll
            for (int j = 0; j < TS_CPW; ++j) {
                double v = __builtin_fma(cur[j].x, t0, cur[j].y * t1);
                for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
                xs[j] = v;
            }
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < TS_CPW; ++j) red2[0][wave * TS_CPW + j] = xs[j];
            }
            __syncthreads();
            if (wave == 0) {
                double *ps = a.pub + (size_t)s * LEAF;
                const double x0 = red2[0][2 * lane], x1 = red2[0][2 * lane + 1];
                publish(ps + 2 * lane, x0);
                publish(ps + 2 * lane + 1, x1);
                bs[2 * lane] = x0;
                bs[2 * lane + 1] = x1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store((gi32 *)(a.state + 1), tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (known < tk + 1) known = tk + 1;
        __syncthreads();                                      // LDS is reused by the next strip
    }
}

}  // namespace

bool trsv_strips_ok(int n, const double *L, size_t ldl)
{
    return n > 0 && n % LEAF == 0 && (ldl & 1) == 0 && (((uintptr_t)L) & 15) == 0;
}

// b (n) := L^-1 b (trans = 0) or L^-T b, n a multiple of 128, L 16-byte aligned with an even leading
// dimension; `state`: 4 ints of device scratch, ZERO on entry (ticket, progress counter, timeout mark);
// `pub`: n doubles of device scratch, every byte 0xFF on entry.  One launch.
int trsv_strips(int n, const double *L, size_t ldl, const double *inv, double *b, int trans, int *state, double *pub,
                hipStream_t st)
{
    if (n <= 0) return 0;
    if (!trsv_strips_ok(n, L, ldl)) { set_error("trsv_strips: shape not supported"); return SGPR_E_ARG; }
    TrsvArgs a{n / LEAF, L, ldl, inv, b, pub, state, trans};
    const int grid = a.T < 256 ? a.T : 256;
    hipLaunchKernelGGL(trsv_strips_kernel, dim3(grid), dim3(TS_T), 0, st, a);
    SGPR_CHECK_LAUNCH();
    return 0;
}

}  // namespace sgpr

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
// published) with agent-scope (sc1, write-through) 8-byte stores; the consumer's lanes re-load their
// elements (sc1) until they are no longer the sentinel -- no flag, no fence, no drain between data and
// signal on the chain of strips.  The progress counter (written behind a drain, off that chain) only tells
// later strips how far they may read without polling.
// The chain itself is one product per strip: the tile whose segment arrives last is not streamed but folded
// into the leaf inverse beforehand (M = inv L(s,s-1), a 128^3 product on the matrix cores), so that the
// arrival of x_{s-1} is followed by  x_s = y - M x_{s-1}  wave by wave (16 rows each), without a workgroup
// barrier; y = inv (b_s - streamed sums) is ready one strip earlier.  n = 16384: 7.4 -> 2.4 us per strip (b, op(inv) and the progress counter are fetched ahead of the chain).
// L itself and the leaf inverses are read-only here: plain 16-byte loads.
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

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
    unsigned long long *dbg;
    // a batch of independent systems of the same order, problem blockIdx.y: element strides of L, inv, b and pub; ints of state
    size_t sL, sInv, sB;
    int sState;
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

#ifdef SGPR_TRSV_DBG
constexpr bool TRSV_DBG = true;
#else
constexpr bool TRSV_DBG = false;   // per-strip time stamps (experiments: make EXTRA=-DSGPR_TRSV_DBG)
#endif
constexpr int SLD = LEAF + 4;               // leading dimension of the staged tile: fragment reads (16 columns x 4 rows) hit every bank twice
typedef double double4_t __attribute__((ext_vector_type(4)));

// element e = 4 jb + r of a lane's 32-vector <-> column 16 jb + 4 r + (lane >> 4): the accumulator layout of
// v_mfma_f64_16x16x4 (row = lane & 15 of the wave's 16 rows)
__device__ __forceinline__ int col_of(int e, int l4) { return 16 * (e >> 2) + 4 * (e & 3) + l4; }
// the same 32 columns per lane group in an order whose addresses in a column-major inv are four immediates per base
__device__ __forceinline__ int colx_of(int e, int l4) { return 16 * (e >> 2) + 4 * l4 + (e & 3); }

template <bool fwd>
__global__ __launch_bounds__(TS_T) void trsv_strips_kernel(const TrsvArgs a_in)
{
    TrsvArgs a = a_in;
    {
        const size_t pb = blockIdx.y;       // (tickets, progress and give-up words are per problem: a workgroup waits for its own system only)
        a.L += pb * a.sL; a.inv += pb * a.sInv; a.b += pb * a.sB; a.pub += pb * a.sB; a.state += pb * a.sState;
    }
    __shared__ double stg[LEAF * SLD];       // the last tile of the strip, staged for the matrix cores; then M, parked
    __shared__ double vec[2][LEAF];          // the segment of the solution a tile is multiplied with
    __shared__ double red[TS_W][LEAF];       // cross-wave partial sums (forward)
    __shared__ double tv[LEAF];              // b_strip - sum
    __shared__ double vecw[TS_W][LEAF];      // the last segment, one copy per wave (no workgroup barrier on the chain)
    __shared__ int sh[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4, i0 = 16 * wave;
    const int T = a.T;
    int known = 0;                           // strips known to be finished (ticket order)
    for (;;) {
        if (tid == 0) { sh[0] = atomicAdd(a.state, 1); sh[1] = 0; }
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
        const double *inv_tile = a.inv + (size_t)s * LEAF * LEAF;
        // op(inv), a row per lane: element (row i0 + l15, column colx_of(e)) of inv (forward) / inv^T
        auto load_invx = [&](double2_t (&reg)[TS_CPW], const double *ib) {
#pragma unroll
            for (int e = 0; e < 2 * TS_CPW; ++e) {
                const int c = colx_of(e, l4);
                const double v = fwd ? ib[(size_t)(i0 + l15) + (size_t)c * LEAF] : ib[(size_t)c + (size_t)(i0 + l15) * LEAF];
                if (e & 1) reg[e >> 1].y = v; else reg[e >> 1].x = v;
            }
        };
        double acc_r0 = 0.0, acc_r1 = 0.0;                    // forward: sums of this lane's two rows
        double acc_c[TS_CPW];                                 // backward: sums of this wave's 16 columns
#pragma unroll
        for (int j = 0; j < TS_CPW; ++j) acc_c[j] = 0.0;
        double2_t cur[TS_CPW], nxt[TS_CPW];
        double seg_next = 0.0;
        bool have_next = false;
        const int ns = tk > 0 ? tk - 1 : 0;                   // tiles that are streamed; the last one becomes M
        // ---- the last tile (the one whose segment arrives last) is folded into the leaf inverse ahead of time:
        //   forward   x_s = inv (b_s - sum_{q < tk-1} L(s,q) x_q) - M x_{s-1},   M = inv L(s, s-1)
        //   backward  x_s = inv^T (b_s - sum ...)                 - M x_{s+1},   M = inv^T L(s+1, s)^T
        // so that the chain of strips carries ONE 128 x 128 product per strip, each wave owning 16 rows of it
        // (no cross-wave sum, no workgroup barrier between the arrival of a segment and the next one's
        // departure).  M is a 128^3 product on the matrix cores: the tile staged in LDS, inv straight from L2.
        if (tk > 0) {
            load_tile(tile_ptr(tk - 1), a.ldl, lane, wave, nxt);
#pragma unroll
            for (int j = 0; j < TS_CPW; ++j)
                *reinterpret_cast<double2_t *>(stg + (wave * TS_CPW + j) * SLD + 2 * lane) = nxt[j];
            __syncthreads();
            double4_t m[8];
#pragma unroll
            for (int jb = 0; jb < 8; ++jb) m[jb] = double4_t{0.0, 0.0, 0.0, 0.0};
            if (fwd) {
                // M[i][j] = sum_k inv[i][k] T[k][j], k <= i
                for (int k0 = 0; k0 < 16 * (wave + 1); k0 += 4) {
                    const double bi = inv_tile[(size_t)(i0 + l15) + (size_t)(k0 + l4) * LEAF];
#pragma unroll
                    for (int jb = 0; jb < 8; ++jb)
                        m[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(stg[(k0 + l4) + (16 * jb + l15) * SLD], bi, m[jb], 0, 0, 0);
                }
            } else {
                // M[i][j] = sum_k inv[k][i] T[j][k], k >= i
                for (int k0 = 16 * wave; k0 < LEAF; k0 += 4) {
                    const double bi = inv_tile[(size_t)(k0 + l4) + (size_t)(i0 + l15) * LEAF];
#pragma unroll
                    for (int jb = 0; jb < 8; ++jb)
                        m[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(stg[(16 * jb + l15) + (k0 + l4) * SLD], bi, m[jb], 0, 0, 0);
                }
            }
            __syncthreads();                                  // every wave is done with the staged tile
#pragma unroll
            for (int jb = 0; jb < 8; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) stg[(4 * jb + r) * TS_T + tid] = m[jb][r];   // parked until the end
        }
        // this thread's element of b is fetched now, not on the chain
        double *bs = a.b + (size_t)s * LEAF;
        const double bpre = fwd ? (tid < LEAF ? bs[tid] : 0.0) : bs[wave * TS_CPW + (lane >> 2)];
        if (ns == 0) load_invx(cur, inv_tile);
        else load_tile(tile_ptr(0), a.ldl, lane, wave, cur);
        for (int q = 0; q < ns; ++q) {
            // the next tile's loads go out before anything that may wait
            // (behind the last streamed tile: op(inv), a row per lane; the base pointer is made opaque so that its 32
            // addresses are formed here and not carried through the loop)
            if (q + 1 < ns) {
                load_tile(tile_ptr(q + 1), a.ldl, lane, wave, nxt);
            } else {
                const double *ib = inv_tile;
                asm volatile("" : "+s"(ib));
                load_invx(nxt, ib);
            }
            // segment q: fetched one step ahead, or (at the frontier) polled for: the 128 lanes that carry
            // the segment poll the counter themselves and load their element the moment it moves
            if (q >= known) {
                if (tid < LEAF) {
                    // how far the others are (later segments up to there need no polling): read BEFORE the poll, so
                    // that this second round trip is not paid between a segment's arrival and its use
                    int prog = 0;
                    if (tid == 0) prog = __hip_atomic_load((gi32 *)(a.state + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned spins = 0;
                    unsigned long long bits;
                    gu64 *src = (gu64 *)(seg_ptr(q) + tid);
                    while ((bits = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == UNPUBLISHED) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (20u << 20)) break;                    // ~ seconds: give up
                    }
                    vec[q & 1][tid] = __longlong_as_double((long long)bits);
                    if (bits == UNPUBLISHED) __hip_atomic_store((gi32 *)(a.state + 2), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (tid == 0) sh[2 + (q & 1)] = bits == UNPUBLISHED ? -1 : prog;
                }
                __syncthreads();
                const int r = sh[2 + (q & 1)];
                if (TRSV_DBG && a.dbg && tid == 0 && q == ns - 1) a.dbg[4 * tk + 3] = __builtin_amdgcn_s_memrealtime();
                if (r < 0) return;
                known = r > q + 1 ? r : q + 1;
            } else {
                const double seg = have_next ? seg_next : (tid < LEAF ? load_sc1(seg_ptr(q) + tid) : 0.0);
                if (tid < LEAF) {
                    // the progress counter said this segment is there; if the sentinel is still what comes back, the
                    // visibility assumption of the fence-free hand-off has failed: report it, do not feed NaNs to alpha
                    if ((unsigned long long)__double_as_longlong(seg) == UNPUBLISHED)
                        __hip_atomic_store((gi32 *)(a.state + 2), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vec[q & 1][tid] = seg;
                }
                __syncthreads();
            }
            // prefetch the following segment when it is known to be there
            have_next = (q + 1 < ns) && (q + 1 < known);
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
        // ---- t = b_s - (streamed sums), still one strip behind the frontier
        if (ns == 0) {
            if (fwd) { if (tid < LEAF) tv[tid] = bpre; }
            else if ((lane & 3) == 0) tv[wave * TS_CPW + (lane >> 2)] = bpre;
        } else if (fwd) {
            red[wave][2 * lane] = acc_r0;
            red[wave][2 * lane + 1] = acc_r1;
            __syncthreads();
            if (tid < LEAF) {
                double r = 0.0;
#pragma unroll
                for (int w = 0; w < TS_W; ++w) r += red[w][tid];
                tv[tid] = bpre - r;
            }
        } else {
            // column sums over the wave's 64 lanes as a butterfly that halves the values a lane carries at every
            // step (8 + 4 + 2 + 1 + 1 + 1 exchanges instead of 16 x 6): lane l ends with column l >> 2
            double v8[8], v4[4], v2[2], v1;
            const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = (b5 ? acc_c[8 + j] : acc_c[j]) + __shfl_xor(b5 ? acc_c[j] : acc_c[8 + j], 32, 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = (b4 ? v8[4 + j] : v8[j]) + __shfl_xor(b4 ? v8[j] : v8[4 + j], 16, 64);
#pragma unroll
            for (int j = 0; j < 2; ++j) v2[j] = (b3 ? v4[2 + j] : v4[j]) + __shfl_xor(b3 ? v4[j] : v4[2 + j], 8, 64);
            v1 = (b2 ? v2[1] : v2[0]) + __shfl_xor(b2 ? v2[0] : v2[1], 4, 64);
            v1 += __shfl_xor(v1, 2, 64);
            v1 += __shfl_xor(v1, 1, 64);
            if ((lane & 3) == 0) tv[wave * TS_CPW + (lane >> 2)] = bpre - v1;
        }
        __syncthreads();
        // y = op(inv) t for this wave's 16 rows (a row per lane): 32 products per lane, then the four lanes of a
        // row add up
        double y = 0.0;
#pragma unroll
        for (int e = 0; e < 2 * TS_CPW; ++e) y = __builtin_fma((e & 1) ? cur[e >> 1].y : cur[e >> 1].x, tv[colx_of(e, l4)], y);
        y += __shfl_xor(y, 16, 64);
        y += __shfl_xor(y, 32, 64);
        if (TRSV_DBG && a.dbg && tid == 0) a.dbg[4 * tk + 0] = __builtin_amdgcn_s_memrealtime();
        if (tk > 0) {
            // ---- the chain: the last segment arrives -> x = y - M seg -> published, wave by wave
            double mm[2 * TS_CPW];
#pragma unroll
            for (int e = 0; e < 2 * TS_CPW; ++e) mm[e] = stg[e * TS_T + tid];
            gu64 *src = (gu64 *)(seg_ptr(tk - 1) + 2 * lane);
            unsigned long long b0, b1 = 0;
            unsigned spins = 0;
            for (;;) {
                b0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b0 != UNPUBLISHED && b1 != UNPUBLISHED) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (20u << 20)) break;                            // ~ seconds: give up
            }
            if (b0 == UNPUBLISHED || b1 == UNPUBLISHED) {
                __hip_atomic_store((gi32 *)(a.state + 2), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh[1] = 1;
            }
            if (TRSV_DBG && a.dbg && tid == 0) a.dbg[4 * tk + 1] = __builtin_amdgcn_s_memrealtime();
            double *vw = vecw[wave];
            *reinterpret_cast<double2_t *>(vw + 2 * lane) =
                double2_t{__longlong_as_double((long long)b0), __longlong_as_double((long long)b1)};
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double p = 0.0;
#pragma unroll
            for (int e = 0; e < 2 * TS_CPW; ++e) p = __builtin_fma(mm[e], vw[col_of(e, l4)], p);
            p += __shfl_xor(p, 16, 64);
            p += __shfl_xor(p, 32, 64);
            y -= p;
        }
        if (l4 == 0) {
            publish(a.pub + (size_t)s * LEAF + i0 + l15, y);
            bs[i0 + l15] = y;
        }
        if (TRSV_DBG && a.dbg && tid == 0) a.dbg[4 * tk + 2] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                      // every wave's part is out; LDS is reused by the next strip
        // (a maximum, not an overwrite: strips finish in dependency order, but nothing here should rest on that)
        if (tid == 0) (void)__hip_atomic_fetch_max((gi32 *)(a.state + 1), tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sh[1]) return;
        if (known < tk + 1) known = tk + 1;
        __syncthreads();
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
    return trsv_strips_batch(1, n, L, 0, ldl, inv, 0, b, 0, trans, state, 0, pub, st);
}

// The same for `nbatch` independent systems of one order in ONE launch (batch.hip: the solves of a batch of mid-size fits):
// problem p has its factor at L + p sL, its leaf inverses at inv + p sInv, its right-hand side at b + p sB, its publication
// buffer at pub + p sB and its state words at state + p sState (zero / all-ones on entry as above).
int trsv_strips_batch(int nbatch, int n, const double *L, size_t sL, size_t ldl, const double *inv, size_t sInv, double *b, size_t sB,
                      int trans, int *state, int sState, double *pub, hipStream_t st)
{
    if (n <= 0 || nbatch <= 0) return 0;
    if (!trsv_strips_ok(n, L, ldl) || (sL & 1) || nbatch > 65535) { set_error("trsv_strips: shape not supported"); return SGPR_E_ARG; }
    TrsvArgs a{n / LEAF, L, ldl, inv, b, pub, state, trans, nullptr, sL, sInv, sB, sState};
    const bool dbg = TRSV_DBG && nbatch == 1 && getenv("SGPR_TRSV_DBG") != nullptr;
    if (dbg) { (void)hipMalloc((void **)&a.dbg, sizeof(unsigned long long) * 4 * a.T); (void)hipMemset(a.dbg, 0, sizeof(unsigned long long) * 4 * a.T); }
    const int grid = a.T < 256 ? a.T : 256;
    if (trans) hipLaunchKernelGGL(trsv_strips_kernel<false>, dim3(grid, nbatch), dim3(TS_T), 0, st, a);
    else       hipLaunchKernelGGL(trsv_strips_kernel<true>, dim3(grid, nbatch), dim3(TS_T), 0, st, a);
    SGPR_CHECK_LAUNCH();
    if (dbg) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(4 * (size_t)a.T);
        (void)hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> hop, vis, cmp, slack, vis2, prep;
        for (int t = 3; t < a.T; ++t) {
            if (h[4 * t + 3]) { vis2.push_back(((double)h[4 * t + 3] - (double)h[4 * (t - 2) + 2]) * 0.01); prep.push_back(((double)h[4 * t + 0] - (double)h[4 * t + 3]) * 0.01); }
            hop.push_back(((double)h[4 * t + 2] - (double)h[4 * (t - 1) + 2]) * 0.01);
            vis.push_back(((double)h[4 * t + 1] - (double)h[4 * (t - 1) + 2]) * 0.01);
            cmp.push_back(((double)h[4 * t + 2] - (double)h[4 * t + 1]) * 0.01);
            slack.push_back(((double)h[4 * (t - 1) + 2] - (double)h[4 * t + 0]) * 0.01);
        }
        auto med = [](std::vector<double> &x) { std::sort(x.begin(), x.end()); return x.empty() ? 0.0 : x[x.size() / 2]; };
        auto q10 = [](std::vector<double> &x) { return x.empty() ? 0.0 : x[x.size() / 10]; };
        fprintf(stderr, "publish(s-2) -> seen in stream (after barrier) med %.2f | -> y ready med %.2f us (%zu polled)\n", med(vis2), med(prep), prep.size());
        fprintf(stderr, "trsv trans=%d T=%d: hop med %.2f us | publish(s-1) -> seen(s) med %.2f | seen -> publish med %.2f | y ready before publish(s-1): med %.2f p10 %.2f us | total %.1f us\n",
                trans, a.T, med(hop), med(vis), med(cmp), med(slack), q10(slack), ((double)h[4 * (a.T - 1) + 2] - (double)h[2]) * 0.01);
        (void)hipFree(a.dbg);
    }
    return 0;
}

}  // namespace sgpr

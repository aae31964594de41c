// trsm.hip -- triangular solves with a BLOCK of right-hand sides,  Y := L^-1 B  /  X := L^-T Y,  as ONE launch each.
//
// Replaces solve_cholesky (python/functions/func.py:174-177 -> LAPACK dtrtrs x2) for several right-hand sides at once:
// the work BASELINE config 05_tokamak names ("multi-RHS predict TRSM"), and what the reference does implicitly with
// matmul(Kyinv, ztrain) per call (python/05_tokamak/SympGPR/sympgpr.f90:72,85,121).
//
// Round 3 solved this by recursion over the MFMA GEMM kernel: ~3000 dependent launches at n = 98304, most of them one
// 64 x 128 tile.  Here it is the design of trsv.hip carried over to 64 columns: every 128-row strip (forward) / 128-column
// strip (backward) is owned by one workgroup that streams its tiles of L ONCE for all 64 right-hand sides (LDS-DMA, double
// buffered, 128 x 32 x 64 products per chunk on the matrix cores), strips are dealt in dependency order by a ticket, and
// the solved 128 x 64 segments are handed from strip to strip through memory behind a progress counter.
//
// Roofline: L is read once per triangular solve (4 n^2 B; 2 n^2 nrhs flop for the pair): at 64 right-hand sides the two
// bounds meet (16 flop/B against a machine balance of ~13), below that the HBM read is the bound.
//
// Layout: the right-hand sides live in a scratch image Y[k][MS_YLD] (row k = row of the system, 64 columns + 16 of
// padding: the rows are the LDS image of the B operand, 2 * 80 mod 64 banks = 32), solved in place: rows of strip s hold
// B_s until strip s publishes Y_s there.
//
// The chain of strips carries ONE product per arriving segment.  With S = B_s - sum_{q < tk-2} op(L_q) Y_q (streamed):
//     Y_s = op(inv) S - M2 Y_{s-2} - M1 Y_{s-1},    M1 = op(inv) op(L(s, s-1)),  M2 = op(inv) op(L(s, s-2))
// M1, M2 (128^3 products) and Z = op(inv) S are formed before their segments arrive; when Y_{s-2} and then Y_{s-1} are
// published the strip only multiplies (128 x 128 x 64, 6.8 us of one CU's matrix cores) and subtracts.
//
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): producer = write-through (sc1) stores of the segment, every
// wave drains (vmcnt(0)), workgroup barrier, one lane raises the progress counter (agent-scope atomic max); consumer = ONE
// relaxed poll by one lane, ONE agent-scope acquire, vmcnt(0), workgroup barrier, then LDS-DMA loads.  Every strip starts
// with such an acquire too (its CU's L1 may hold lines of rows that have been published since), and B_s itself is read
// with sc1 loads (never allocated in L1).  A wait is always for a SMALLER ticket, whatever the residency; every wait is
// bounded in real time and a give-up is reported through state[2] (-> SGPR_E_HIP), never a hang.
#include "common.h"
#include <cstdlib>

namespace sgpr {

namespace {

constexpr int MS_T = 512;                    // 8 waves: 4 (32-row blocks) x 2 (32-column blocks) of the 128 x 64 result
constexpr int MS_NC = 64;                    // right-hand sides per pass
constexpr int MS_BK = 32;                    // reduction depth of one staged chunk
constexpr int MS_YLD = TRSM_YLD;             // 80: row stride of the right-hand-side image (global AND LDS)
constexpr int AN_LD = LEAF + 16;             // "N" image of an A chunk: [k][144]  (rows contiguous in memory)
constexpr int XT_LD = MS_BK + 2;             // "T" image of a chunk whose reduction index is contiguous in memory: [row][34]
constexpr int A_ELEMS = MS_BK * AN_LD;       // 4608 doubles (>= 128 * 34 = 4352)
constexpr int B_ELEMS = MS_BK * MS_YLD;      // 2560 doubles (>= 64 * 34 = 2176)
constexpr int STAGE_ELEMS = A_ELEMS + B_ELEMS;
constexpr int WG_SCRATCH = TRSM_WG_SCRATCH;  // per workgroup: M1^T, M2^T (128 x 128 each), S (128 x 80)
static_assert(WG_SCRATCH == 2 * LEAF * LEAF + LEAF * MS_YLD, "scratch layout");

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) int gi32;

struct TrsmArgs {
    int T;                 // strips (n = 128 T)
    const double *L;
    size_t ldl;
    const double *inv;     // leaf inverses, LEAF x LEAF each
    double *Y;             // n x MS_YLD image: right-hand sides in, solution out
    double *scratch;       // gridDim.x * WG_SCRATCH doubles
    int *state;            // [0] ticket, [1] ready (strips published, in ticket order), [2] give-up flag
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

__device__ __forceinline__ void dma16(const double *src, double *lds_dst)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// One operand of a 128 (or 64) x 32 chunk product.  `p` points at element (row 0, reduction index 0) of the operand,
// `ld` is its leading dimension in memory.
//   N: the non-reduction index is contiguous in memory (element (row, k) at row + k ld)
//   T: the reduction index is contiguous in memory     (element (row, k) at k + row ld)
struct Operand { const double *p; unsigned ld; };

// ---- LDS-DMA staging of chunk kc (reduction indices [32 kc, 32 kc + 32)).  A wave instruction moves 64 granules of 16 B
// to 1 KiB of consecutive LDS; which granule a lane fetches is free, so the padded images are filled in image order.
template <bool AT>
__device__ __forceinline__ void issue_A(const Operand &A, int kc, double *As, int wave, int lane)
{
    if constexpr (!AT) {
        // image [k][144]: instruction j = reduction index k, lanes = 128 consecutive rows
        const double *src = A.p + (size_t)(MS_BK * kc) * A.ld + 2 * lane;
#pragma unroll
        for (int x = 0; x < MS_BK / 8; ++x) {
            const int j = wave + 8 * x;
            dma16(src + (size_t)j * A.ld, As + j * AN_LD);
        }
    } else {
        // image [row][34]: granule g = 17 row + rp (rp = 16: the pad, fetched from a valid dummy address)
        const double *src = A.p + MS_BK * kc;
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            const int j = wave + 8 * x;
            if (j < (LEAF * 17) / 64) {
                const int g = 64 * j + lane, row = g / 17, rp = g - 17 * row;
                dma16(src + (size_t)row * A.ld + 2 * (rp < 16 ? rp : 15), As + 128 * j);
            }
        }
    }
}
template <bool BT>
__device__ __forceinline__ void issue_B(const Operand &B, int kc, double *Bs, int wave, int lane)
{
    if constexpr (!BT) {
        // image [k][80]: granule g = 40 k + jp (jp >= 32: pad)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const int j = wave + 8 * x;
            if (j < (MS_BK * MS_YLD) / 128) {
                const int g = 64 * j + lane, k = g / 40, jp = g - 40 * k;
                dma16(B.p + (size_t)(MS_BK * kc + k) * B.ld + 2 * (jp < 32 ? jp : 31), Bs + 128 * j);
            }
        }
    } else {
        // image [col][34], 64 columns
        const double *src = B.p + MS_BK * kc;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const int j = wave + 8 * x;
            if (j < (MS_NC * 17) / 64) {
                const int g = 64 * j + lane, col = g / 17, rp = g - 17 * col;
                dma16(src + (size_t)col * B.ld + 2 * (rp < 16 ? rp : 15), Bs + 128 * j);
            }
        }
    }
}

// acc (this wave's 32 x 32 block of the 128 x 64 result, 2 x 2 accumulators of v_mfma_f64_16x16x4) += A chunk . B chunk
// accumulator layout: acc[x][y][r] = element (row 32 wm + 16 x + 4 r + (lane >> 4), column 32 wn + 16 y + (lane & 15))
template <bool AT, bool BT>
__device__ __forceinline__ void compute_chunk(double4_t (&acc)[2][2], const double *As, const double *Bs, int wm, int wn, int l15,
                                              int l4)
{
    const double *pa = AT ? As + (32 * wm + l15) * XT_LD + l4 : As + l4 * AN_LD + 32 * wm + l15;
    const double *pb = BT ? Bs + (32 * wn + l15) * XT_LD + l4 : Bs + l4 * MS_YLD + 32 * wn + l15;
#pragma unroll
    for (int kk = 0; kk < MS_BK / 4; ++kk) {
        const double a0 = AT ? pa[4 * kk] : pa[4 * kk * AN_LD];
        const double a1 = AT ? pa[4 * kk + 16 * XT_LD] : pa[4 * kk * AN_LD + 16];
        const double b0 = BT ? pb[4 * kk] : pb[4 * kk * MS_YLD];
        const double b1 = BT ? pb[4 * kk + 16 * XT_LD] : pb[4 * kk * MS_YLD + 16];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

struct Ctl {
    int *state;
    int *sh;               // 2 ints of LDS
    int known;             // strips known to be published (ticket order)
    int tid;
};

constexpr unsigned long long WAIT_LIMIT_TICKS = 500000000ull;   // 5 s of the 100 MHz real-time counter

// Blocks until `need` strips are published.  All threads call it; returns false when the wait was given up.
__device__ __forceinline__ bool wait_ready(Ctl &c, int need)
{
    if (c.known >= need) return true;
    if (c.tid == 0) {
        int v;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned it = 0;
        bool ok = true;
        while ((v = __hip_atomic_load((gi32 *)(c.state + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
            __builtin_amdgcn_s_sleep(2);
            if ((++it & 63u) == 0) {
                if (__hip_atomic_load((gi32 *)(c.state + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                    __builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) { ok = false; break; }
            }
        }
        if (!ok) __hip_atomic_store((gi32 *)(c.state + 2), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        c.sh[0] = ok ? v : -1;
    }
    __syncthreads();
    const int v = c.sh[0];
    __syncthreads();
    if (v < 0) return false;
    c.known = v;
    return true;
}

// acc += sum over tiles q = 0 .. ntiles-1 of A_q (128 x 128) . B_q (128 x 64), chunk by chunk, the copy of chunk t + 1
// in flight under the products of chunk t, across tile boundaries.  `tile(q, A, B)` names the operands of product q;
// with POLL, B_q is segment q of the chain and may be read only once q + 1 strips are published.
template <bool AT, bool BT, bool POLL, class TileFn>
__device__ __forceinline__ bool stream_products(double4_t (&acc)[2][2], int ntiles, TileFn &&tile, Ctl &c, double *smem)
{
    const int tid = c.tid, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2, l15 = lane & 15, l4 = lane >> 4;
    const int nch = (LEAF / MS_BK) * ntiles;
    bool primed = false;
    Operand A{}, B{};
    auto issue = [&](int t) {
        const int q = t >> 2, kc = t & 3;
        if (kc == 0) tile(q, A, B);
        double *As = smem + (t & 1) * STAGE_ELEMS, *Bs = As + A_ELEMS;
        issue_A<AT>(A, kc, As, wave, lane);
        issue_B<BT>(B, kc, Bs, wave, lane);
    };
    for (int t = 0; t < nch; ++t) {
        if (!primed) {
            if (POLL && (t & 3) == 0 && !wait_ready(c, (t >> 2) + 1)) return false;
            issue(t);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const bool nxt = t + 1 < nch && (!POLL || ((t + 1) & 3) != 0 || ((t + 1) >> 2) < c.known);
        if (nxt) issue(t + 1);
        const double *As = smem + (t & 1) * STAGE_ELEMS;
        compute_chunk<AT, BT>(acc, As, As + A_ELEMS, wm, wn, l15, l4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        primed = nxt;
    }
    return true;
}

__device__ __forceinline__ void zero_acc(double4_t (&acc)[2][2])
{
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = double4_t{0.0, 0.0, 0.0, 0.0};
}

template <bool fwd>
__global__ __launch_bounds__(MS_T) void trsm_strips_kernel(const TrsmArgs a)
{
    __shared__ double smem[2 * STAGE_ELEMS];
    __shared__ int sh[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2, l15 = lane & 15, l4 = lane >> 4;
    const int T = a.T;
    double *const Mt1 = a.scratch + (size_t)blockIdx.x * WG_SCRATCH;   // M1 stored row-major (= M1^T column-major)
    double *const Mt2 = Mt1 + LEAF * LEAF;
    double *const Ss = Mt2 + LEAF * LEAF;                              // S, [k][80]
    Ctl c{a.state, sh, 0, tid};
    constexpr bool AT = !fwd;            // backward: every tile of L and the leaf inverse act transposed
    // element e = (x, y, r) of this lane: row i(x, r), column j(y)
    auto row_of = [&](int x, int r) { return 32 * wm + 16 * x + 4 * r + l4; };
    auto col_of = [&](int y) { return 32 * wn + 16 * y + l15; };
    for (;;) {
        if (tid == 0) {
            sh[2] = atomicAdd(a.state, 1);
            // what is published by now may be read without polling; the acquire also drops every line this CU's L1 holds of
            // rows that have been published since it read them
            sh[3] = __hip_atomic_load((gi32 *)(a.state + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const int tk = sh[2];
        c.known = sh[3];
        __syncthreads();
        if (tk >= T) return;
        const int s = fwd ? tk : T - 1 - tk;                 // this workgroup's strip
        // dependency q = 0 .. tk-1 of the strip, in the order the segments are published:
        //   forward: tile (s, q), segment q;  backward: tile (T-1-q, s), segment T-1-q
        auto tile_ptr = [&](int q) {
            return fwd ? a.L + (size_t)s * LEAF + (size_t)q * LEAF * a.ldl
                       : a.L + (size_t)(T - 1 - q) * LEAF + (size_t)s * LEAF * a.ldl;
        };
        auto seg_ptr = [&](int q) { return a.Y + (size_t)(fwd ? q : T - 1 - q) * LEAF * MS_YLD; };
        const double *inv_s = a.inv + (size_t)s * LEAF * LEAF;
        const int nfold = tk < 2 ? tk : 2, ns = tk - nfold;
        double4_t acc[2][2];

        // ---- M1 = op(inv) op(tile_{tk-1}), M2 = op(inv) op(tile_{tk-2}), 64 columns per pass, stored row-major
        for (int f = 0; f < nfold; ++f) {
            const double *tl = tile_ptr(tk - 1 - f);
            double *Mt = f ? Mt2 : Mt1;
            for (int pass = 0; pass < 2; ++pass) {
                zero_acc(acc);
                // forward: B[red j][col k] = tile[j + k ldl] (reduction index contiguous: T image);
                // backward: B[red j][col k] = tile[k + j ldl] (N image with the tile's leading dimension)
                auto one = [&](int, Operand &A, Operand &B) {
                    A = Operand{inv_s, (unsigned)LEAF};
                    B = fwd ? Operand{tl + (size_t)(64 * pass) * a.ldl, (unsigned)a.ldl} : Operand{tl + 64 * pass, (unsigned)a.ldl};
                };
                (void)stream_products<AT, fwd, false>(acc, 1, one, c, smem);
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Mt[row_of(x, r) * LEAF + 64 * pass + col_of(y)] = acc[x][y][r];
            }
        }
        // ---- streamed part: acc = sum_{q < ns} op(tile_q) Y_q
        zero_acc(acc);
        if (ns > 0) {
            auto tl = [&](int q, Operand &A, Operand &B) {
                A = Operand{tile_ptr(q), (unsigned)a.ldl};
                B = Operand{seg_ptr(q), (unsigned)MS_YLD};
            };
            if (!stream_products<AT, false, true>(acc, ns, tl, c, smem)) return;
        }
        // ---- S = B_s - acc  ->  scratch;  Z = op(inv) S
        double *ys = a.Y + (size_t)s * LEAF * MS_YLD;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = row_of(x, r) * MS_YLD + col_of(y);
                    Ss[o] = load_sc1(ys + o) - acc[x][y][r];
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        double4_t z[2][2];
        zero_acc(z);
        {
            auto one = [&](int, Operand &A, Operand &B) {
                A = Operand{inv_s, (unsigned)LEAF};
                B = Operand{Ss, (unsigned)MS_YLD};
            };
            (void)stream_products<AT, false, false>(z, 1, one, c, smem);
        }
        // ---- the chain: Y_{s-2} arrives -> Z -= M2 Y_{s-2};  Y_{s-1} arrives -> Z -= M1 Y_{s-1}
        for (int f = nfold - 1; f >= 0; --f) {
            const int q = tk - 1 - f;
            if (!wait_ready(c, q + 1)) return;
            zero_acc(acc);
            auto one = [&](int, Operand &A, Operand &B) {
                A = Operand{f ? Mt2 : Mt1, (unsigned)LEAF};      // row-major M: its reduction index is contiguous
                B = Operand{seg_ptr(q), (unsigned)MS_YLD};
            };
            (void)stream_products<true, false, false>(acc, 1, one, c, smem);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) z[x][y] -= acc[x][y];
        }
        // ---- publish
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) store_sc1(ys + row_of(x, r) * MS_YLD + col_of(y), z[x][y][r]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) (void)__hip_atomic_fetch_max((gi32 *)(a.state + 1), tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (this barrier is not decoration: without it hipcc folds the one-lane region above and the one-lane ticket draw at the
        // top of the loop into an exit of an inner loop that the other 511 threads keep running -- with the OLD ticket)
        __syncthreads();
    }
}

// B (n x nc, column-major) <-> the image Y[k][80]; columns nc .. 63 of the image are zero
__global__ __launch_bounds__(256) void pack_rhs_kernel(int n, int nc, const double *B, size_t ldb, double *Y)
{
    __shared__ double tile[64][65];
    const int k0 = blockIdx.x * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) tile[j][tx] = (j < nc && k0 + tx < n) ? B[(size_t)(k0 + tx) + (size_t)j * ldb] : 0.0;
    __syncthreads();
    for (int k = ty; k < 64; k += 4)
        if (k0 + k < n) Y[(size_t)(k0 + k) * MS_YLD + tx] = tile[tx][k];
}
__global__ __launch_bounds__(256) void unpack_rhs_kernel(int n, int nc, const double *Y, double *B, size_t ldb)
{
    __shared__ double tile[64][65];
    const int k0 = blockIdx.x * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4) tile[tx][k] = (k0 + k < n) ? Y[(size_t)(k0 + k) * MS_YLD + tx] : 0.0;
    __syncthreads();
    for (int j = ty; j < nc; j += 4)
        if (k0 + tx < n) B[(size_t)(k0 + tx) + (size_t)j * ldb] = tile[j][tx];
}

}  // namespace

bool trsm_strips_ok(int n, const double *L, size_t ldl)
{
    static const bool off = [] { const char *e = getenv("SGPR_TRSM"); return e && e[0] == 'r'; }();
    return !off && n >= 2 * LEAF && n % LEAF == 0 && (ldl & 1) == 0 && (((uintptr_t)L) & 15) == 0 &&
           (size_t)LEAF * ldl < ((size_t)1 << 31);
}

int trsm_strips_grid(int n) { const int T = n / LEAF; return T < 256 ? T : 256; }

size_t trsm_strips_scratch(int n)
{
    return ((size_t)n * MS_YLD + (size_t)trsm_strips_grid(n) * WG_SCRATCH) * sizeof(double);
}

// B (n x nrhs, column-major, device) := L^-T L^-1 B, 64 columns per pass through the image; `state`: 8 ints of device
// scratch (two solves); `scratch`: trsm_strips_scratch(n) bytes.
int potrs_strips(int n, const double *L, size_t ldl, const double *inv, double *B, size_t ldb, int nrhs, int *state,
                 double *scratch, hipStream_t st)
{
    if (n <= 0 || nrhs <= 0) return 0;
    if (!trsm_strips_ok(n, L, ldl)) { set_error("potrs_strips: shape not supported"); return SGPR_E_ARG; }
    const int T = n / LEAF, grid = trsm_strips_grid(n);
    double *Y = scratch, *wg = scratch + (size_t)n * MS_YLD;
    for (int c0 = 0; c0 < nrhs; c0 += MS_NC) {
        const int nc = nrhs - c0 < MS_NC ? nrhs - c0 : MS_NC;
        hipLaunchKernelGGL(pack_rhs_kernel, dim3((n + 63) / 64), dim3(256), 0, st, n, nc, B + (size_t)c0 * ldb, ldb, Y);
        SGPR_CHECK_LAUNCH();
        SGPR_HIP(hipMemsetAsync(state, 0, 8 * sizeof(int), st));
        TrsmArgs a{T, L, ldl, inv, Y, wg, state, 0};
        hipLaunchKernelGGL(trsm_strips_kernel<true>, dim3(grid), dim3(MS_T), 0, st, a);
        SGPR_CHECK_LAUNCH();
        a.state = state + 4;
        a.trans = 1;
        hipLaunchKernelGGL(trsm_strips_kernel<false>, dim3(grid), dim3(MS_T), 0, st, a);
        SGPR_CHECK_LAUNCH();
        hipLaunchKernelGGL(unpack_rhs_kernel, dim3((n + 63) / 64), dim3(256), 0, st, n, nc, Y, B + (size_t)c0 * ldb, ldb);
        SGPR_CHECK_LAUNCH();
    }
    return 0;
}

}  // namespace sgpr

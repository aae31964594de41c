// trsm.hip -- triangular solves with a BLOCK of right-hand sides,  Y := L^-1 B  /  X := L^-T Y,  as ONE launch each.
//
// Replaces solve_cholesky (python/functions/func.py:174-177 -> LAPACK dtrtrs x2) for several right-hand sides at once:
// the work BASELINE config 05_tokamak names ("multi-RHS predict TRSM"), and what the reference does implicitly with
// matmul(Kyinv, ztrain) per call (python/05_tokamak/SympGPR/sympgpr.f90:72,85,121).
//
// Round 3 solved this by recursion over the MFMA GEMM kernel: ~3000 dependent launches at n = 98304, most of them one
// 64 x 128 tile (n = 16384, 64 right-hand sides: 44.8 ms).  Here one launch per triangular solve, two classes of workgroups:
//
//   STREAM class (most of the chip): strip s (128 rows forward / 128 columns backward) is one task (long strips: several, see
//   piece_of), dealt in dependency order by a ticket.  It streams the tiles of L of its strip ONCE for all 64 right-hand sides (LDS-DMA into a 3-deep ring
//   for L and a 2-deep ring for the solved segments, 128 x 32 x 64 products per chunk on the matrix cores) up to F tiles short
//   of the diagonal, and hands S = B_s - sum_{q < tk-F} op(L_q) Y_q to the chain.  Before that it folds the F tiles next to the
//   diagonal into the leaf inverse, M_f = op(inv_s) op(L(s, s-+f)) (128^3 products, off the chain).
//
//   CHAIN class (a few dozen workgroups): task (s, c) carries columns 16 c .. 16 c + 15 of strip s through the recurrence
//       Y_s = op(inv_s) S - sum_{f = F..1} M_f Y_{s-+f}
//   as 128 x 128 x 16 products straight out of registers: a wave owns 16 rows, its operand fragments of M_f come from a
//   fragment-ordered scratch image (16-byte loads), the segment it multiplies with is polled for element by element.
//   The chain of strips therefore carries 32 MFMAs (~1.7 us) and one memory round trip per strip instead of a 128 x 128 x 64
//   product behind a flag, an acquire and a workgroup barrier (first version of this file: 17 us per strip).
//
// Roofline: L is read once per triangular solve (4 n^2 B; 2 n^2 nrhs flop for the pair): at 64 right-hand sides the two
// bounds meet (16 flop/B against a machine balance of ~13), below that the HBM read is the bound.
//
// Layout: right-hand sides and solution are images [k][MS_YLD] (row k = row of the system, 64 columns + 16 of padding: the
// rows are the LDS image of the B operand, 2 * 80 mod 64 banks = 32).  Three images: the input B, the published solution P
// and the hand-over S; P and S start as all-ones bit patterns.
//
// Hand-offs (MI355X_MICROARCH.md, inter-workgroup visibility):
//   * chain <- chain, chain <- stream (S), piece <- piece (running partial sums): the data is its own signal.  8-byte
//     write-through (sc1) stores of values that are never the all-ones pattern (NaNs are canonicalised), consumers re-load their
//     elements (sc1) until none is the pattern ("8-byte granules"; the form trsv.hip uses).
//   * chain <- stream (op(inv) and the folded tiles M_f): stored (sc1), drained, barrier, one relaxed store to the strip's flag.
//   * stream <- chain (bulk reads of published segments by LDS-DMA): per quarter a progress counter, raised behind drain +
//     workgroup barrier; the reader polls the four counters once (one lane), ONE agent-scope acquire, vmcnt(0), barrier.
// Every stream task also starts with an acquire.  Forward progress: chain workgroups have the lowest block ids (dispatched
// first); a chain task waits for older chain tasks and for its own strip's stream task, a stream task for older chain tasks:
// the oldest unfinished task of either class can always finish.  Every wait is bounded in real time; a give-up is reported
// through state[2] (-> SGPR_E_HIP), never a hang.
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace sgpr {

namespace {

constexpr int MS_T = 512;                    // 8 waves
constexpr int MS_NC = 64;                    // right-hand sides per pass
#ifndef SGPR_TRSM_BK
#define SGPR_TRSM_BK 32
#endif
#ifndef SGPR_TRSM_NT
#define SGPR_TRSM_NT 1
#endif
constexpr bool TRSM_NT = SGPR_TRSM_NT != 0;
constexpr int MS_BK = SGPR_TRSM_BK;          // reduction depth of one staged chunk: 16 or 32
constexpr int CPT = LEAF / MS_BK, CPT_SH = MS_BK == 16 ? 3 : 2; // chunks per tile
static_assert((MS_BK == 16 || MS_BK == 32) && CPT == 1 << CPT_SH, "chunks per tile");
constexpr int MS_YLD = TRSM_YLD;             // 80: row stride of the images (global AND LDS)
constexpr int MS_F = TRSM_FOLD;              // tiles next to the diagonal that are folded into the leaf inverse
constexpr int XT_LD = MS_BK + 2;             // "T" image of a chunk whose reduction index is contiguous in memory: [row][MS_BK + 2]
constexpr int NPAD = 16;                     // row pad of the "N" images: [k][ROWS + 16], rows 32 banks apart -- the four k-rows of a
                                             // fragment read fall on disjoint banks per half wave.  (A pad of 8 -- two-way conflicts on
                                             // half the lanes -- measured the same: LDS is a quarter busy here.)
// ring slots, in doubles: the images ("T" [128][MS_BK + 2] / "N" [MS_BK][144] for L, "N" [MS_BK][80] / "T" [64][MS_BK + 2] for the
// segments) rounded up to whole wave instructions per issuing wave
constexpr int A_ELEMS = MS_BK == 16 ? 2304 : 4608;
constexpr int B_ELEMS = MS_BK == 16 ? 1280 : 2560;
// What bounds the products of a stream task on a quiet chip turned out to be INSTRUCTION ISSUE, not memory: one wave issues at
// most one instruction per ~4 cycles, and the first forms of this loop spent 340 of them per 16 MFMAs, most on index arithmetic
// (10-12 us per tile whatever the ring depth, the pad or the barrier form).  Hence running pointers (Copier), role-specialised
// copies of the loop, constant vmcnt immediates, the copies issued under the MFMAs, the last MFMA group of a chunk deferred
// behind the barrier.  With the whole chip streaming the latency of the copies counts as well: thin chunks and a deep ring for
// L (six chunks = 104 KB in flight per CU).
// The two operands are issued by DIFFERENT waves: vmcnt retires in order, so a wave that waits for the next chunk of the segment
// would wait for every older copy of L too.
constexpr int A_STAGES = MS_BK == 16 ? 7 : 3, B_STAGES = MS_BK == 16 ? 3 : 2;
constexpr int A_W0 = 0, A_NW = 6, B_W0 = 6, B_NW = 2;   // waves 0..5 issue the copies of L, waves 6, 7 those of the segment
constexpr int MFRAG = LEAF * LEAF;           // doubles of one folded tile M_f
constexpr int MSLOTS = TRSM_FOLD + 1;        // fragment-ordered operands of a chain task per strip: op(inv), M_1 .. M_F

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) int gi32;

struct TrsmArgs {
    int T;                 // strips (n = 128 T)
    int nchain;            // workgroups of the chain class: blockIdx.x < nchain
    const double *L;
    size_t ldl;
    const double *inv;     // leaf inverses, LEAF x LEAF each
    const double *Bin;     // image of the right-hand sides
    double *P;             // image of the solution, all-ones on entry
    double *S;             // image of the hand-over stream -> chain, all-ones on entry
    double *M;             // T * MSLOTS * MFRAG doubles, fragment order: per strip op(inv) (slot 0) and the folded tiles M_1 .. M_F
    int *mflag;            // per strip: complete slots (zero on entry)
    double *X;             // partial sums of the strips that are streamed in several pieces, XPART doubles each, all-ones on entry
    int piece;             // most tiles one stream task takes (piece_of)
    int ncols;             // right-hand sides of this pass (1 .. 64): waves whose 32 columns are all padding skip their products
    int *state;            // [0] stream ticket, [1] chain ticket, [2] give-up flag
    int *ready;            // [c] strips whose quarter c is published (ticket order)
    unsigned long long *dbg;   // debug builds: 16 time stamps (100 MHz) per strip, or null
    int fake_b;                // debug builds, experiment: the streamed products read one cache-hot segment chunk over and over
};

#ifdef SGPR_TRSM_DBG
constexpr bool TRSM_DBG = true;
#else
constexpr bool TRSM_DBG = false;   // per-strip time stamps (experiments: make EXTRA=-DSGPR_TRSM_DBG, SGPR_TRSM_DBG=1)
#endif

constexpr int DBGW = 32;                      // debug builds: stamp words per strip
constexpr unsigned long long UNPUBLISHED = ~0ull;

// A stream ticket = one PIECE of a strip's streamed tiles, or the FOLD of one of its tiles.  Strip tk streams ns = tk - MS_F
// tiles; dealt whole, the last tickets are the longest tasks of the launch (n = 98304: 7.7 ms each of 17) and the chip idles
// behind them for half of that on average.  So a strip with more than C tiles goes out as np = ceil(ns / C) tickets over equal
// shares of its tiles, oldest segments first: the first np - 1 (helpers) pass a running 128 x 64 partial sum along through X
// (piece p adds piece p - 1's), the last one (the owner: the youngest segments, op(inv) for the chain and the hand-over S) takes
// the sum of all before it.  The folds M_f = op(inv) op(tile), f = 1 .. min(tk, F), are tickets of their own between the helpers
// and the owner: five of them in a row were 144 us at the head of every launch -- a third of a solve at n = 4096 -- during which
// the rest of the chip had nothing to do.  Tickets stay in dependency order: a helper waits only for segments older than the
// owner's, a fold for nothing, the owner for its helpers (smaller tickets).
struct Piece { int tk, p, np, x0, fold; };   // strip, piece, pieces, slot of the strip's first partial sum in X; fold: 0 or f
constexpr size_t XPART = (size_t)LEAF * 64;
// helper pieces (= partial sums) of the strips before tk: sum over ns = 1 .. tk - F - 1 of floor((ns - 1) / C), in closed form
__host__ __device__ inline long pieces_before(int tk, int C)
{
    const long N = (long)tk - TRSM_FOLD - 1;
    if (N <= 0) return 0;
    const long a = N / C, b = N % C;
    return (long)C * a * (a - 1) / 2 + a * b;
}
// tickets of the strips before tk: one owner each, the helper pieces, min(j, F) folds
__host__ __device__ inline long tickets_before(int tk, int C)
{
    const long folds = tk <= TRSM_FOLD ? (long)tk * (tk - 1) / 2 : (long)TRSM_FOLD * (TRSM_FOLD - 1) / 2 + (long)TRSM_FOLD * (tk - TRSM_FOLD);
    return tk + pieces_before(tk, C) + folds;
}
// ticket u of a solve with T strips (tk = T: past the end)
__host__ __device__ inline Piece piece_of(int u, int C, int T)
{
    int lo = 0, hi = T;                       // the largest tk with tickets_before(tk) <= u
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tickets_before(mid, C) <= u) lo = mid; else hi = mid - 1;
    }
    const int tk = lo;
    if (tk >= T) return Piece{T, 0, 1, 0, 0};
    const int v = u - (int)tickets_before(tk, C), ns = tk > TRSM_FOLD ? tk - TRSM_FOLD : 0, np = ns > C ? (ns + C - 1) / C : 1;
    const int nfold = tk < TRSM_FOLD ? tk : TRSM_FOLD, x0 = (int)pieces_before(tk, C);
    if (v < np - 1) return Piece{tk, v, np, x0, 0};
    if (v < np - 1 + nfold) return Piece{tk, 0, np, x0, v - (np - 1) + 1};
    return Piece{tk, np - 1, np, x0, 0};
}
// stream tickets / partial sums of a solve with T strips (host: sizes the grid and the scratch)
inline size_t piece_tickets(int T, int C) { return (size_t)tickets_before(T, C); }
inline size_t piece_partials(int T, int C) { return (size_t)pieces_before(T, C); }
constexpr unsigned long long WAIT_LIMIT_TICKS = 500000000ull;   // 5 s of the 100 MHz real-time counter

__device__ __forceinline__ void store_sc1(double *p, double v)
{
    __hip_atomic_store((gu64 *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish(double *p, double v)
{
    if (v != v) v = __longlong_as_double(0x7FF8000000000000ll);       // never the all-ones pattern
    store_sc1(p, v);
}
__device__ __forceinline__ unsigned long long load_bits_sc1(const double *p)
{
    return __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// NT: non-temporal (the factor's tiles are read exactly once, by one CU: streamed past the caches so that the solved segments,
// which every later strip reads again, stay in the Infinity Cache)
template <bool NT>
__device__ __forceinline__ void dma16(const double *src, double *lds_dst)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, NT ? 2 : 0);
}

// ---- LDS-DMA staging.  One operand of a (ROWS x 32) chunk product; `p` points at element (row 0, reduction index 0):
//   N: the non-reduction index is contiguous in memory (element (row, k) at row + k ld)  -> image [k][ROWS + 16]
//   T: the reduction index is contiguous in memory     (element (row, k) at k + row ld)  -> image [row][34]
// A wave instruction moves 64 granules of 16 B to 1 KiB of consecutive LDS; which granule a lane fetches is free, so the
// padded images are filled in image order (pad granules re-fetch a neighbour).  The per-lane offsets depend on the
// leading dimension only and are formed once per task.
// The copy stream of one operand as one lane of an issuing wave sees it: running source pointers (one per wave instruction of a
// chunk), advanced by a constant per chunk and by another constant at a tile boundary -- the tiles of a strip and the segments
// they multiply are linear streams, so nothing is recomputed inside the product loop (an earlier form spent ~340 instructions
// per chunk, most of them index arithmetic: at one instruction per ~4 cycles per wave that was as long as the 16 MFMAs).
template <bool TR, int ROWS, int W0, int NW, int STAGES, int ELEMS, bool NT>
struct Copier {
    static constexpr int GPR = TR ? MS_BK / 2 + 1 : (ROWS + NPAD) / 2;       // granules per image row
    static constexpr int NJ = (TR ? ROWS * GPR : MS_BK * GPR) / 64;          // wave instructions per chunk
    static constexpr int NX = (NJ + NW - 1) / NW;                            // per issuing wave: EVERY issuing wave makes NX copies
    static_assert((TR ? ROWS * GPR : MS_BK * GPR) % 64 == 0, "image is a whole number of wave instructions");
    static_assert(128 * NW * NX <= ELEMS, "the padding copies land inside the ring slot");
    const double *ptr[NX];
    long step, adj;          // elements per chunk; extra elements at a tile boundary
    double *dst;             // this wave's first destination in ring slot 0
    int slot, left;          // ring slot of the next chunk, chunks left in its tile
    // `p0`: element (row 0, reduction index 0) of the first chunk; ld: leading dimension of the operand in memory.
    // Called by the issuing waves only.  (A list that is not a multiple of the issuing waves is padded: the extra copy re-fetches
    // a granule into the unused tail of the slot, so that the counted waits are the same immediate for every wave of a role.)
    __device__ __forceinline__ void init(const double *p0, unsigned ld, long step_, long adj_, double *ring, int wave, int lane)
    {
        step = step_; adj = adj_; slot = 0; left = CPT;
        const int w = wave - W0;
        dst = ring + 128 * w;
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            int g = 64 * (w + NW * x) + lane;
            if (w + NW * x >= NJ) g = lane;                                  // padding copy
            const int r = g / GPR, c = g - GPR * r;
            const int cmax = TR ? MS_BK / 2 - 1 : ROWS / 2 - 1;
            ptr[x] = p0 + ((size_t)r * ld + 2u * (unsigned)(c < cmax ? c : cmax));
        }
    }
    __device__ __forceinline__ void issue()
    {
        double *d = dst + slot * ELEMS;
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            dma16<NT>(ptr[x], d + 128 * NW * x);
            ptr[x] += step;
        }
        if (--left == 0) {
            left = CPT;
#pragma unroll
            for (int x = 0; x < NX; ++x) ptr[x] += adj;
        }
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
};

__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p;
}
template <int OFF>
__device__ __forceinline__ double ds_read_f64(unsigned addr)
{
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
#define TRSM_LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// acc (this wave's 32 x 32 block of the 128 x 64 result, 2 x 2 accumulators of v_mfma_f64_16x16x4) += A chunk . B chunk
// accumulator layout: acc[x][y][r] = element (row 32 wm + 16 x + 4 r + (lane >> 4), column 32 wn + 16 y + (lane & 15))
// The fragment reads are explicit ds_read_b64 with counted waits, the reads of k-step kk + 1 in flight under the MFMAs of
// k-step kk (left to itself hipcc sinks every read next to its use: read, lgkmcnt(0), four MFMAs, read, ...).
template <bool AT, bool BT, int KK>
__device__ __forceinline__ void read_frags(unsigned aA, unsigned aB, double (&f)[4])
{
    constexpr int AN = LEAF + NPAD, BN = MS_NC + NPAD;
    f[0] = ds_read_f64<(AT ? 4 * KK : 4 * KK * AN) * 8>(aA);
    f[1] = ds_read_f64<(AT ? 4 * KK + 16 * XT_LD : 4 * KK * AN + 16) * 8>(aA);
    f[2] = ds_read_f64<(BT ? 4 * KK : 4 * KK * BN) * 8>(aB);
    f[3] = ds_read_f64<(BT ? 4 * KK + 16 * XT_LD : 4 * KK * BN + 16) * 8>(aB);
}
__device__ __forceinline__ void mfma4(double4_t (&acc)[2][2], const double (&f)[4])
{
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0], f[2], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0], f[3], acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[1], f[2], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[1], f[3], acc[1][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}
// One chunk: MS_BK / 4 k-steps.  The MFMAs of the LAST k-step are not issued here: they are left to the next call (or to the
// caller, behind the loop), which first asks for its own first fragments -- so the matrix cores have work while the first LDS
// reads behind the barrier are in flight.  `pend`: fragments of the previous chunk's last k-step (valid when `have`).
template <bool AT, bool BT, int KK, int NK>
struct ChunkSteps {
    // k-steps KK .. NK-1, the fragments of k-step KK - 1 in `cur`: reads of KK go out, then the MFMAs of KK - 1
    template <class Mid>
    static __device__ __forceinline__ void run(double4_t (&acc)[2][2], unsigned aA, unsigned aB, double (&cur)[4], double (&pend)[4], Mid &&mid)
    {
        if constexpr (KK == NK - 1) {
            read_frags<AT, BT, KK>(aA, aB, pend); TRSM_LGKM_WAIT(4); mfma4(acc, cur);
            if constexpr (KK == 1) { mid(); __builtin_amdgcn_sched_barrier(0); }
        } else {
            double nxt[4];
            read_frags<AT, BT, KK>(aA, aB, nxt); TRSM_LGKM_WAIT(4); mfma4(acc, cur);
            if constexpr (KK == 1) { mid(); __builtin_amdgcn_sched_barrier(0); }     // the copies of the chunks ahead go out under these MFMAs
            ChunkSteps<AT, BT, KK + 1, NK>::run(acc, aA, aB, nxt, pend, mid);
        }
    }
};
template <bool AT, bool BT, class Mid>
__device__ __forceinline__ void compute_chunk(double4_t (&acc)[2][2], const double *As, const double *Bs, int wm, int wn, int l15,
                                              int l4, double (&pend)[4], bool have, Mid &&mid, bool skip = false)
{
    constexpr int AN = LEAF + NPAD, BN = MS_NC + NPAD;
    if (skip) { mid(); __builtin_amdgcn_sched_barrier(0); return; }      // (wave-uniform; the copies of the chunks ahead still go out)
    const unsigned aA = lds_addr(AT ? As + (32 * wm + l15) * XT_LD + l4 : As + l4 * AN + 32 * wm + l15);
    const unsigned aB = lds_addr(BT ? Bs + (32 * wn + l15) * XT_LD + l4 : Bs + l4 * BN + 32 * wn + l15);
    double f0[4];
    __builtin_amdgcn_sched_barrier(0);
    read_frags<AT, BT, 0>(aA, aB, f0);
    __builtin_amdgcn_sched_barrier(0);
    if (have) mfma4(acc, pend);
    ChunkSteps<AT, BT, 1, MS_BK / 4>::run(acc, aA, aB, f0, pend, mid);
    TRSM_LGKM_WAIT(0);                         // (the slot may be refilled behind the next barrier)
}

// s_waitcnt vmcnt(n) for a run-time n (the counter is an immediate)
__device__ __forceinline__ void wait_vmcnt(int n)
{
#define TRSM_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
    TRSM_VM(1) TRSM_VM(2) TRSM_VM(3) TRSM_VM(4) TRSM_VM(5) TRSM_VM(6) TRSM_VM(7) TRSM_VM(8) TRSM_VM(9) TRSM_VM(10)
    TRSM_VM(11) TRSM_VM(12) TRSM_VM(13) TRSM_VM(14) TRSM_VM(15) TRSM_VM(16) TRSM_VM(17) TRSM_VM(18) TRSM_VM(19) TRSM_VM(20)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef TRSM_VM
}

struct Ctl {
    int *state, *ready;
    int *sh;               // 2 ints of LDS
    int known;             // strips known to be published in all four quarters (ticket order)
    int qbase;             // the polled streams: dependency index of the first tile of the piece
    bool skip;             // this wave's 32 columns of the streamed products are padding (fewer than 33 right-hand sides): no MFMAs
    int tid;
    unsigned long long *dbg_wait;   // debug builds: [0] time of the first wait that had to poll, [1] the tile it was for
};

__device__ __forceinline__ bool gave_up(const int *state)
{
    return __hip_atomic_load((gi32 *)(state + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
__device__ __forceinline__ void give_up(int *state)
{
    __hip_atomic_store((gi32 *)(state + 2), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Blocks until `need` strips are published (all quarters).  All threads call it; false when the wait was given up.
__device__ __forceinline__ bool wait_ready(Ctl &c, int need)
{
    if (c.known >= need) return true;
    if (c.tid == 0) {
        int v;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (TRSM_DBG && c.dbg_wait && c.dbg_wait[0] == 0) { c.dbg_wait[0] = t0; c.dbg_wait[1] = (unsigned long long)(need - c.qbase); }
        unsigned it = 0;
        bool ok = true;
        for (;;) {
            const int v0 = __hip_atomic_load((gi32 *)(c.ready + 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int v1 = __hip_atomic_load((gi32 *)(c.ready + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int v2 = __hip_atomic_load((gi32 *)(c.ready + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int v3 = __hip_atomic_load((gi32 *)(c.ready + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = min(min(v0, v1), min(v2, v3));
            if (v >= need) break;
            __builtin_amdgcn_s_sleep(2);
            if ((++it & 63u) == 0 && (gave_up(c.state) || __builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS)) { ok = false; break; }
        }
        if (!ok) give_up(c.state);
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

// One operand stream of stream_products: first element, leading dimension, elements per chunk and extra elements per tile
struct OpStream { const double *p0; unsigned ld; long step, adj; };

// acc += sum over tiles q = 0 .. ntiles-1 of A_q (128 x 128) . B_q (128 x 64), chunk by chunk: the copies of A two chunks, those
// of B one chunk ahead of the products, across tile boundaries.  With POLL, B_q is segment q of the chain and may be read
// only once q + 1 strips are published (A, the factor itself, is read ahead regardless).
// ROLE 0: this wave issues the copies of A, ROLE 1: those of B -- two copies of the loop, so that nothing inside it asks which.
template <bool AT, bool BT, bool POLL, int ROLE>
__device__ __forceinline__ bool stream_loop(double4_t (&acc)[2][2], int ntiles, const OpStream &oa, const OpStream &ob, Ctl &c,
                                            double *smem)
{
    const int tid = c.tid, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2, l15 = lane & 15, l4 = lane >> 4;
    const int nch = CPT * ntiles;
    double *const Aring = smem, *const Bring = smem + A_STAGES * A_ELEMS;
    typedef Copier<AT, LEAF, A_W0, A_NW, A_STAGES, A_ELEMS, POLL && TRSM_NT> CA;      // POLL: the streamed tiles of L
    typedef Copier<BT, MS_NC, B_W0, B_NW, B_STAGES, B_ELEMS, false> CB;
    CA ca;
    CB cb;
    if (ROLE == 0) ca.init(oa.p0, oa.ld, oa.step, oa.adj, Aring, wave, lane);
    else           cb.init(ob.p0, ob.ld, ob.step, ob.adj, Bring, wave, lane);
    constexpr int NPER = ROLE == 0 ? CA::NX : CB::NX;        // copies per chunk of this wave
    constexpr int LEAD = ROLE == 0 ? A_STAGES - 2 : B_STAGES - 2;   // chunks of this wave's operand in flight behind chunk t + 1
    static_assert(MS_BK == 16 ? (CA::NX * (A_STAGES - 2) == 15 && CB::NX * (B_STAGES - 2) == 5)
                              : (CA::NX * (A_STAGES - 2) == 6 && CB::NX * (B_STAGES - 2) == 0), "the steady-state vmcnt immediates below");
    int ia = 0, ib = 0;                                      // next chunk of A / of B to be asked for
    int blimit = POLL ? min(nch, CPT * (c.known - c.qbase)) : nch;   // chunks of B that may be asked for
    for (; ia < nch && ia < A_STAGES - 1; ++ia)
        if (ROLE == 0) ca.issue();
    int sa_slot = 0, sb_slot = 0;                            // ring slots of chunk t
    double pend[4] = {0.0, 0.0, 0.0, 0.0};                   // fragments of the last k-step of the chunk before (compute_chunk)
    for (int t = 0; t < nch; ++t) {
        if (ib <= t) {
            // chunk t's segment has not even been asked for (start, or the frontier): wait until it may be
            if (POLL && !wait_ready(c, c.qbase + (t >> CPT_SH) + 1)) return false;
            if (POLL) blimit = min(nch, CPT * (c.known - c.qbase));
            if (ROLE == 1) cb.issue();
            ib = t + 1;
        }
        // chunk t has landed (this wave's share: younger copies stay in flight), and every wave is done with chunk t - 1
        {
            const int ahead = (ROLE == 0 ? ia : ib) - (t + 1);
            if (ahead == LEAD) {
                if (MS_BK == 16) {
                    if (ROLE == 0) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                    else           asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                } else {
                    if (ROLE == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else           asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                wait_vmcnt(NPER * ahead);
            }
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        compute_chunk<AT, BT>(acc, Aring + sa_slot * A_ELEMS, Bring + sb_slot * B_ELEMS, wm, wn, l15, l4, pend, t > 0, [&]() {
            // ... whose ring slots take the next copies
            if (ia < nch) { if (ROLE == 0) ca.issue(); ++ia; }
            const int lim = min(blimit, t + B_STAGES);
            while (ib < lim) { if (ROLE == 1) cb.issue(); ++ib; }
        }, POLL && c.skip);
        sa_slot = sa_slot + 1 == A_STAGES ? 0 : sa_slot + 1;
        sb_slot = sb_slot + 1 == B_STAGES ? 0 : sb_slot + 1;
    }
    if (nch > 0 && !(POLL && c.skip)) mfma4(acc, pend);
    // every wave is done with the last chunk before anybody refills the rings
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    return true;
}
template <bool AT, bool BT, bool POLL>
__device__ __forceinline__ bool stream_products(double4_t (&acc)[2][2], int ntiles, const OpStream &oa, const OpStream &ob, Ctl &c,
                                                double *smem)
{
    const int wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
    return wave < A_NW ? stream_loop<AT, BT, POLL, 0>(acc, ntiles, oa, ob, c, smem)
                       : stream_loop<AT, BT, POLL, 1>(acc, ntiles, oa, ob, c, smem);
}

__device__ __forceinline__ void zero_acc(double4_t (&acc)[2][2])
{
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = double4_t{0.0, 0.0, 0.0, 0.0};
}

// position of element (i, k) of a folded tile in its fragment-ordered image: wave i / 16 reads its operand fragments of
// k-steps 2 p, 2 p + 1 with ONE 16-byte load per lane (lane = (k & 3) * 16 + (i & 15))
__device__ __forceinline__ int frag_index(int i, int k)
{
    const int w = i >> 4, kk = k >> 2, lane = (k & 3) * 16 + (i & 15);
    return ((w * 16 + (kk >> 1)) * 64 + lane) * 2 + (kk & 1);
}

// ---------------------------------------------------------------------------------------------------- stream class
template <bool fwd>
__device__ __forceinline__ bool stream_task(const TrsmArgs &a, const Piece pc, Ctl &c, double *smem)
{
    const int tk = pc.tk;
    const bool owner = !pc.fold && pc.p == pc.np - 1;
    const int tid = c.tid, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2, l15 = lane & 15, l4 = lane >> 4;
    const int T = a.T;
    constexpr bool AT = !fwd;            // backward: every tile of L and the leaf inverse act transposed
    auto row_of = [&](int x, int r) { return 32 * wm + 16 * x + 4 * r + l4; };
    auto col_of = [&](int y) { return 32 * wn + 16 * y + l15; };
    const int s = fwd ? tk : T - 1 - tk;                 // this task's strip
    // dependency q = 0 .. tk-1 of the strip, in the order the segments are published:
    //   forward: tile (s, q), segment q;  backward: tile (T-1-q, s), segment T-1-q
    auto tile_ptr = [&](int q) {
        return fwd ? a.L + (size_t)s * LEAF + (size_t)q * LEAF * a.ldl
                   : a.L + (size_t)(T - 1 - q) * LEAF + (size_t)s * LEAF * a.ldl;
    };
    auto seg_ptr = [&](int q) { return a.P + (size_t)(fwd ? q : T - 1 - q) * LEAF * MS_YLD; };
    const double *inv_s = a.inv + (size_t)s * LEAF * LEAF;
    const int ns = tk - (tk < MS_F ? tk : MS_F);
    const int q0 = (int)((long)pc.p * ns / pc.np), q1 = (int)((long)(pc.p + 1) * ns / pc.np);   // this piece's dependencies
    auto stamp = [&](int i) {
        if (TRSM_DBG && a.dbg && tid == 0 && owner) {
            if (i == 0) a.dbg[DBGW * tk + 12] = (unsigned long long)(q1 - q0);
            a.dbg[DBGW * tk + i] = __builtin_amdgcn_s_memrealtime();
            if (i == 1 || i == 2) a.dbg[DBGW * tk + 9 + i] = __builtin_amdgcn_s_memtime();   // shader clock of the streamed part
        }
    };
    stamp(0);
    c.dbg_wait = (TRSM_DBG && a.dbg && owner) ? a.dbg + DBGW * tk + 8 : nullptr;
    c.qbase = q0;
    double4_t acc[2][2];
    // ---- a fold ticket: M_f = op(inv) op(tile_{tk-f}), 64 columns per pass, stored in fragment order (write-through); then the
    // strip's count of complete slots goes up by one (every store drained, barrier, one relaxed add)
    if (pc.fold) {
        const int f = pc.fold;
        const double *tl = tile_ptr(tk - f);
        double *Mf = a.M + ((size_t)s * MSLOTS + f) * MFRAG;
        for (int pass = 0; pass < 2; ++pass) {
            zero_acc(acc);
            // A = op(inv): forward N image (rows contiguous), backward T image (the inverse acts transposed).
            // forward: B[red j][col k] = tile[j + k ldl] (reduction index contiguous: T image);
            // backward: B[red j][col k] = tile[k + j ldl] (N image with the tile's leading dimension)
            const OpStream oa{inv_s, (unsigned)LEAF, fwd ? (long)MS_BK * LEAF : (long)MS_BK, 0};
            const OpStream ob{fwd ? tl + (size_t)(64 * pass) * a.ldl : tl + 64 * pass, (unsigned)a.ldl,
                              fwd ? (long)MS_BK : (long)MS_BK * (long)a.ldl, 0};
            (void)stream_products<AT, fwd, false>(acc, 1, oa, ob, c, smem);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r) store_sc1(Mf + frag_index(row_of(x, r), 64 * pass + col_of(y)), acc[x][y][r]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) (void)__hip_atomic_fetch_add((gi32 *)(a.mflag + s), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    if (owner) {
        // slot 0: op(inv_s) itself in fragment order (the chain multiplies S with it like any M_f), counted like a fold: the chain
        // task starts on its products, with the segments that are already there, once the strip's min(tk, F) + 1 slots are complete
        // -- long before S is
        double *M0 = a.M + (size_t)s * MSLOTS * MFRAG;
#pragma unroll 4
        for (int e = 0; e < LEAF * LEAF / MS_T; ++e) {
            const int idx = tid + MS_T * e, i = idx & (LEAF - 1), k = idx >> 7;
            store_sc1(M0 + frag_index(i, k), fwd ? inv_s[i + LEAF * k] : inv_s[k + LEAF * i]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) (void)__hip_atomic_fetch_add((gi32 *)(a.mflag + s), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp(1);
    // ---- streamed part: acc = sum_{q < ns} op(tile_q) Y_q.  Both operands are linear streams: forward, tile (s, q + 1) follows
    // tile (s, q) 128 columns on and segment q + 1 follows segment q; backward, tile (T-2-q, s) and segment T-2-q lie 128 rows
    // BEFORE their predecessors (256 rows back from where the eighth chunk ended).
    // The owner's sums start at -B_s (fetched here, under the stream): S = -(sum - B_s) then goes out the moment the last
    // product is done -- fetching B_s behind the stream held the hand-over back by ~5 us, on the cycle that bounds the launch at
    // orders where the chain does (stream tail + chain task, spread over F + 1 strips).
    zero_acc(acc);
    if (owner) {
        const double *bs = a.Bin + (size_t)s * LEAF * MS_YLD;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[x][y][r] = -bs[row_of(x, r) * MS_YLD + col_of(y)];
    }
    if (q1 > q0) {
        const OpStream oa{tile_ptr(q0), (unsigned)a.ldl, fwd ? (long)MS_BK * (long)a.ldl : (long)MS_BK, fwd ? 0L : -2L * LEAF};
        OpStream ob{seg_ptr(q0), (unsigned)MS_YLD, (long)MS_BK * MS_YLD, fwd ? 0L : -2L * LEAF * MS_YLD};
        if (TRSM_DBG && a.fake_b == 1) { ob.p0 = a.P + (size_t)(fwd ? 0 : T - 1) * LEAF * MS_YLD; ob.step = 0; ob.adj = 0; }   // experiment: every segment chunk from ONE cache-hot place (wrong results)
        c.skip = 32 * wn >= a.ncols;
        const bool ok = stream_products<AT, false, true>(acc, q1 - q0, oa, ob, c, smem);
        c.skip = false;
        if (!ok) return false;
    }
    stamp(2);
    // ---- RUNNING partial sums: piece p adds what piece p - 1 left (the sum of pieces 0 .. p - 1; value v of thread tid at
    // [v][tid], 8-byte granules re-loaded until none is the all-ones pattern) and, unless it is the owner, leaves the new sum for
    // piece p + 1.  The owner therefore takes ONE partial sum on the way to S, however many pieces the strip has (gathering all
    // of them itself cost 1 us each there, on the cycle that bounds chain-bound launches); a helper's wait is for a piece that
    // covers older segments and started earlier.
    if (pc.p > 0) {
        const double *xs = a.X + (size_t)(pc.x0 + pc.p - 1) * XPART + tid;
        unsigned long long bits[16];
        unsigned long long t0 = 0;
        unsigned it = 0;
        bool ok = true;
        for (;;) {
            bool missing = false;
#pragma unroll
            for (int v = 0; v < 16; ++v) { bits[v] = load_bits_sc1(xs + (size_t)v * MS_T); missing |= bits[v] == UNPUBLISHED; }
            if (!__builtin_amdgcn_ballot_w64(missing)) break;
            if (it == 0) t0 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_sleep(8);
            if ((++it & 63u) == 0 && (gave_up(a.state) || __builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS)) { ok = false; break; }
        }
        if (!__syncthreads_and(ok)) { if (tid == 0) give_up(a.state); return false; }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[x][y][r] += __longlong_as_double((long long)bits[(x * 2 + y) * 4 + r]);
    }
    if (!owner) {
        double *xs = a.X + (size_t)(pc.x0 + pc.p) * XPART + tid;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) publish(xs + (size_t)((x * 2 + y) * 4 + r) * MS_T, acc[x][y][r]);
        return true;
    }
    // ---- S = -acc = B_s - sums -> the hand-over image.  Every M_f store of this workgroup has been drained by now (each wave waits
    // for vmcnt(0) in front of every barrier of the products above; explicitly once more here): S is the chain's signal.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double *ss = a.S + (size_t)s * LEAF * MS_YLD;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = row_of(x, r) * MS_YLD + col_of(y);
                publish(ss + o, -acc[x][y][r]);
            }
    stamp(3);
    return true;
}

// ---------------------------------------------------------------------------------------------------- chain class
// acc += A (this wave's 16 rows, fragments af[kk] = A[16 w + l15][4 kk + l4]) . B (128 x 16: one quarter of a segment).
// `seg` = &image[first row of the segment][16 c].  Every wave fetches ITS 16 rows of the quarter (4 elements per lane, the
// ones wave w of the producer published), polling until none is the all-ones pattern, and the eight waves share them through
// LDS behind one barrier: a wave that pulls the whole 16 KB quarter by itself -- the first form -- pays ~4 us of its own memory
// queue per product (MI355X_MICROARCH.md, handoff-payload), 2 KB cost one round trip.
// In three parts, so that the round trip for the NEXT product's rows runs under this product's MFMAs (a chain task is F + 1
// products in a row, and its length is on the cycle that bounds the launch at chain-bound orders):
//   quarter_ask (loads only) ... quarter_have (waits, asks again until the rows are there) ... quarter_multiply.
struct QuarterRows { unsigned long long bits[4]; };
__device__ __forceinline__ void quarter_ask(QuarterRows &q, const double *seg, int wave, int l15, int l4)
{
    const double *src = seg + (size_t)(16 * wave + l4) * MS_YLD + l15;          // element r: row 16 w + 4 r + l4
#pragma unroll
    for (int r = 0; r < 4; ++r) q.bits[r] = load_bits_sc1(src + (size_t)(4 * r) * MS_YLD);
}
__device__ __forceinline__ bool quarter_have(QuarterRows &q, const double *seg, int *state, int wave, int l15, int l4)
{
    unsigned long long t0 = 0;
    unsigned it = 0;
    for (;;) {
        const bool missing = q.bits[0] == UNPUBLISHED || q.bits[1] == UNPUBLISHED || q.bits[2] == UNPUBLISHED || q.bits[3] == UNPUBLISHED;
        if (!__builtin_amdgcn_ballot_w64(missing)) return true;
        if (it == 0) t0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 127u) == 0 && (gave_up(state) || __builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS)) return false;
        quarter_ask(q, seg, wave, l15, l4);
    }
}
// `under`: called once every wave's rows are in LDS and the first fragment reads are out -- the place for the next product's loads
template <typename Under>
__device__ __forceinline__ void quarter_multiply(double4_t &acc, const double (&af)[32], const QuarterRows &q, double *qbuf, int wave,
                                                 int l15, int l4, Under &&under, unsigned long long *st = nullptr)
{
    if (TRSM_DBG && st) st[0] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int r = 0; r < 4; ++r) qbuf[(16 * wave + 4 * r + l4) * 16 + l15] = __longlong_as_double((long long)q.bits[r]);
    // (a bare barrier behind the LDS writes: __syncthreads() would also wait for the fragments of the next M_f, which are
    // meant to arrive under this product)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (TRSM_DBG && st) st[1] = __builtin_amdgcn_s_memrealtime();
    // fragments b[kk] = B[4 kk + l4][l15]: rows of 16 doubles, two consecutive rows cover the 64 banks
    const unsigned ab = lds_addr(qbuf + l4 * 16 + l15);
    double b[32];
    double4_t e = double4_t{0.0, 0.0, 0.0, 0.0}, o = double4_t{0.0, 0.0, 0.0, 0.0};
    __builtin_amdgcn_sched_barrier(0);
#define TRSM_RB(kk) b[kk] = ds_read_f64<(kk) * 4 * 16 * 8>(ab)
#define TRSM_RB8(k0) TRSM_RB(k0); TRSM_RB(k0 + 1); TRSM_RB(k0 + 2); TRSM_RB(k0 + 3); TRSM_RB(k0 + 4); TRSM_RB(k0 + 5); TRSM_RB(k0 + 6); TRSM_RB(k0 + 7)
#define TRSM_MM8(k0)                                                                                                    \
    _Pragma("unroll") for (int kk = k0; kk < k0 + 8; kk += 2) {                                                         \
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], b[kk], e, 0, 0, 0);                                            \
        o = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk + 1], b[kk + 1], o, 0, 0, 0);                                    \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0)
    // eight reads ahead of the MFMAs (the LDS counter holds 15)
    TRSM_RB8(0);
    under();
    __builtin_amdgcn_sched_barrier(0);
    TRSM_RB8(8);  TRSM_LGKM_WAIT(8); TRSM_MM8(0);
    TRSM_RB8(16); TRSM_LGKM_WAIT(8); TRSM_MM8(8);
    TRSM_RB8(24); TRSM_LGKM_WAIT(8); TRSM_MM8(16);
    TRSM_LGKM_WAIT(0); TRSM_MM8(24);
#undef TRSM_RB
#undef TRSM_RB8
#undef TRSM_MM8
    acc += e + o;
    if (TRSM_DBG && st) st[2] = __builtin_amdgcn_s_memrealtime();
}

template <bool fwd>
__device__ __forceinline__ void chain_task(const TrsmArgs &a, int u, int tid, double *smem)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int T = a.T, tk = u >> 2, cq = u & 3;
    const int s = fwd ? tk : T - 1 - tk;
    const int nfold = tk < MS_F ? tk : MS_F;
    auto stamp = [&](int i) { if (TRSM_DBG && a.dbg && tid == 0 && cq == 0) a.dbg[DBGW * tk + i] = __builtin_amdgcn_s_memrealtime(); };
    // two LDS images of a quarter segment, used in turn: product n + 2 writes the image product n read, and every wave has
    // passed the barrier of product n + 1 -- behind its reads of product n -- by then
    double *const qb0 = smem, *const qb1 = smem + LEAF * 16;
    bool ok = true;
    // Y_s = op(inv_s) S - sum_{f = F..1} M_f Y_{s-+f} as nfold + 1 products, ordered by when their right-hand operands arrive:
    //   M_F Y_{s-+F}, ..., M_2 Y_{s-+2}   (segments published two and more steps ago)
    //   op(inv_s) S                       (S: the stream task's hand-over, F + 1 steps behind the chain)
    //   M_1 Y_{s-+1}                      (the predecessor: the chain itself)
    // -- with S first, as this began, every task sat waiting for S and THEN had all F + 1 products in front of it, on the cycle
    // (stream tail + chain task, over F + 1 strips) that bounds the launch at chain-bound orders.
    const int nops = nfold + 1, pos_inv = nfold > 1 ? nfold - 1 : 0;
    auto slot_of = [&](int j) { return j < pos_inv ? nfold - j : (j == pos_inv ? 0 : 1); };
    auto seg_of = [&](int j) {
        const int f = slot_of(j);
        if (f == 0) return (const double *)(a.S + (size_t)s * LEAF * MS_YLD + 16 * cq);
        const int q = tk - f;                                  // the segment published q-th
        return (const double *)(a.P + (size_t)(fwd ? q : T - 1 - q) * LEAF * MS_YLD + 16 * cq);
    };
    // the strip's slots are complete?  (counted up by the strip's fold tickets and its owner; every wave looks itself)
    {
        unsigned long long t0 = 0;
        unsigned it = 0;
        while (__hip_atomic_load((gi32 *)(a.mflag + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nfold + 1) {
            if (it == 0) t0 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_sleep(8);
            if ((++it & 63u) == 0 && (gave_up(a.state) || __builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS)) { ok = false; break; }
        }
    }
    // fragments of the operands, one product ahead (plain 16-byte loads: these lines were stored write-through and drained
    // before the flag went out, and this CU has never read them)
    double af[32];
    double2_t an[16];
    auto load_m = [&](int f) {
        const double *Mf = a.M + ((size_t)((TRSM_DBG && a.fake_b == 2) ? 0 : s) * MSLOTS + f) * MFRAG + ((size_t)(wave * 16) * 64 + lane) * 2;   // (experiment 2: every strip reads strip 0's slots -- cache-hot, wrong results)
#pragma unroll
        for (int p = 0; p < 16; ++p) an[p] = *reinterpret_cast<const double2_t *>(Mf + (size_t)p * 128);
    };
    QuarterRows rows, next;
    double4_t z = double4_t{0.0, 0.0, 0.0, 0.0};
    if (ok) load_m(slot_of(0));
    quarter_ask(next, seg_of(0), wave, l15, l4);
    for (int j = 0; j < nops; ++j) {
#pragma unroll
        for (int p = 0; p < 16; ++p) { af[2 * p] = an[p].x; af[2 * p + 1] = an[p].y; }
        if (j + 1 < nops) load_m(slot_of(j + 1));
        ok &= quarter_have(next, seg_of(j), a.state, wave, l15, l4);
        rows = next;
        double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
        quarter_multiply(acc, af, rows, (j & 1) ? qb1 : qb0, wave, l15, l4, [&]() { if (j + 1 < nops) quarter_ask(next, seg_of(j + 1), wave, l15, l4); },
                         (TRSM_DBG && a.dbg && tid == 0 && cq == 0 && j == nops - 1) ? a.dbg + DBGW * tk + 13 : nullptr);
        if (j == pos_inv) { z += acc; stamp(4); }
        else              z -= acc;
    }
    stamp(5);
    // publish: element (row 16 w + 4 r + l4, column 16 cq + l15)
    double *ps = a.P + (size_t)s * LEAF * MS_YLD + (size_t)(16 * wave + l4) * MS_YLD + 16 * cq + l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) publish(ps + (size_t)(4 * r) * MS_YLD, z[r]);
    stamp(6);
    if (!ok) give_up(a.state);
    // the quarter's progress counter (bulk readers): behind every wave's drain
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) (void)__hip_atomic_fetch_max((gi32 *)(a.ready + cq), tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stamp(7);
}

template <bool fwd>
__global__ __launch_bounds__(MS_T) void trsm_strips_kernel(const TrsmArgs a)
{
    __shared__ double smem[A_STAGES * A_ELEMS + B_STAGES * B_ELEMS];
    __shared__ int sh[8];
    const int tid = threadIdx.x;
    const bool chain = (int)blockIdx.x < a.nchain;
    const int ntasks = chain ? 4 * a.T : a.T;
    Ctl c{a.state, a.ready, sh, 0, 0, false, tid, nullptr};
    for (;;) {
        if (tid == 0) {
            const int u = atomicAdd(a.state + (chain ? 1 : 0), 1);
            sh[2] = u;
            if (!chain) { const Piece pc = piece_of(u, a.piece, a.T); sh[2] = pc.tk; sh[4] = pc.p; sh[5] = pc.np; sh[6] = pc.x0; sh[7] = pc.fold; }
            // what is published by now may be read without polling; the acquire also drops every line this CU's L1 holds of
            // rows that have been published since it read them
            int v = 0x7fffffff;
            for (int q = 0; q < 4; ++q) v = min(v, __hip_atomic_load((gi32 *)(a.ready + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            sh[3] = gave_up(a.state) ? -1 : v;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const int tk = sh[2];
        c.known = sh[3];
        const Piece pc{tk, sh[4], sh[5], sh[6], sh[7]};
        __syncthreads();
        if (tk >= ntasks || c.known < 0) return;
        if (chain) {
            chain_task<fwd>(a, tk, tid, smem);
        } else {
            if (!stream_task<fwd>(a, pc, c, smem)) return;
        }
        // (this barrier is not decoration: without one behind the one-lane regions at the end of a task hipcc folds them and the
        // one-lane ticket draw at the top of the loop into an exit of an inner loop that the other 511 threads keep running --
        // with the OLD ticket)
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

static void dbg_report(const TrsmArgs &a, const char *what, hipStream_t st)
{
    (void)hipStreamSynchronize(st);
    const int T = a.T;
    std::vector<unsigned long long> h(DBGW * (size_t)T);
    (void)hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemset(a.dbg, 0, h.size() * 8);
    auto us = [&](int t, int i, int t2, int j) { return ((double)h[DBGW * t + i] - (double)h[DBGW * t2 + j]) * 0.01; };
    auto med = [](std::vector<double> &x) { std::sort(x.begin(), x.end()); return x.empty() ? 0.0 : x[x.size() / 2]; };
    std::vector<double> step, prep, tile, freetile, spub, slead, zdone, hop, last, mhz, c0, c1, c2;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 8; ++i)
            if (h[DBGW * t + i]) { tmin = std::min(tmin, h[DBGW * t + i]); tmax = std::max(tmax, h[DBGW * t + i]); }
    int nwaited = 0;
    for (int t = MS_F + 2; t < T; ++t) {
        step.push_back(us(t, 6, t - 1, 6));                  // publish to publish (quarter 0)
        prep.push_back(us(t, 1, t, 0));
        const double ntile = (double)h[DBGW * t + 12];
        tile.push_back(us(t, 2, t, 1) / ntile);
        if (h[DBGW * t + 8] && h[DBGW * t + 9] > 16) { freetile.push_back(us(t, 8, t, 1) / (double)(h[DBGW * t + 9] - 1)); ++nwaited; }
        else if (!h[DBGW * t + 8] && ntile > 16) freetile.push_back(us(t, 2, t, 1) / ntile);
        spub.push_back(us(t, 3, t, 2));
        if (h[DBGW * t + 2] > h[DBGW * t + 1] + 2000) mhz.push_back((double)(h[DBGW * t + 11] - h[DBGW * t + 10]) / (double)(h[DBGW * t + 2] - h[DBGW * t + 1]) * 100.0);
        slead.push_back(us(t - 1, 6, t, 3));                 // S handed over how long before the predecessor published
        zdone.push_back(us(t - 1, 6, t, 4));                 // Z formed how long before the predecessor published
        hop.push_back(us(t, 5, t - 1, 6));                   // predecessor published -> last product done here
        c0.push_back(us(t, 13, t - 1, 6));                   // ... -> wave 0 has its rows of the predecessor's segment
        c1.push_back(us(t, 14, t, 13));                      // ... -> every wave has, LDS image complete
        c2.push_back(us(t, 15, t, 14));
                      // ... -> fragments read, 32 MFMAs done
        last.push_back(us(t, 6, t, 5));                      // publish
    }
    fprintf(stderr, "trsm %s T=%d: step %.2f us | (F = %d) owner: slot 0 %.1f us | streamed tile %.2f us each, %.2f before the first wait (%d strips waited) | S out %.1f | "
            "S handed over %.1f us, Z formed %.1f us before the predecessor published | pred. published -> last product done %.2f | publish %.2f | whole launch %.1f us | shader clock %.0f MHz | last product: published -> wave 0 has it %.2f, -> all waves %.2f, -> MFMAs done %.2f\n",
            what, T, med(step), MS_F, med(prep), med(tile), med(freetile), nwaited, med(spub), med(slead), med(zdone), med(hop), med(last),
            (double)(tmax - tmin) * 0.01, med(mhz), med(c0), med(c1), med(c2));
}

}  // namespace

// host view of the stream tickets (libsympgpr_probe.so, tests/test_boundary_cpu.py): ticket u of a solve whose pieces take at
// most C tiles -> (strip, piece, pieces, slot of the strip's first partial sum, fold or 0); counts for T strips -> (tickets, partial sums)
void trsm_piece_of(int u, int C, int T, int out[5])
{
    const Piece p = piece_of(u, C, T);
    out[0] = p.tk; out[1] = p.p; out[2] = p.np; out[3] = p.x0; out[4] = p.fold;
}
void trsm_piece_counts(int T, int C, size_t out[2]) { out[0] = piece_tickets(T, C); out[1] = piece_partials(T, C); }

// every hand-off word of a forward + backward pair back to zero EXCEPT the two give-up words
static __global__ void state_reset_kernel(int *state, int force_giveup)
{
    const int i = threadIdx.x;
    if (i < TRSM_STATE_INTS && i != 2 && i != 6) state[i] = 0;
    if (force_giveup && i == 2) state[2] = 1;
}

bool trsm_strips_ok(int n, const double *L, size_t ldl)
{
    static const bool off = [] { const char *e = getenv("SGPR_TRSM"); return e && e[0] == 'r'; }();
    return !off && n >= 2 * LEAF && n % LEAF == 0 && (ldl & 1) == 0 && (((uintptr_t)L) & 15) == 0 &&
           (size_t)LEAF * ldl < ((size_t)1 << 31);
}

// most tiles of L one stream task takes (piece_of); tunable "trsm_piece" (sgpr_probe_tune)
// Default: 128 (n = 65 536 ... 98 304: 64 ... 256 within 3 %); T / 16, at least 4, up to 256 strips, where every stream task sits at
// the chain's frontier for its whole life: what a task lags per frontier tile adds up over its tiles, on the cycle that bounds
// the launch, and shorter pieces also put the idle part of the chip to work (measured with the running sums, ms at n = 8192 /
// 16 384 / 32 768: 4 tiles 1.01 / 2.11 / 7.5, 8 tiles 1.10 / 1.95 / 5.3, 16 tiles 1.18 / 2.13 / 4.68, 32 tiles 1.26 / 2.29 / 4.66).
static int piece_cap(int T)
{
    const int v = (int)tune("trsm_piece", 0);
    return v <= 0 ? (T <= 256 ? (T / 16 < 4 ? 4 : T / 16) : 128) : (v < 4 ? 4 : v);
}

// bytes of scratch: three images, the folded tiles and the partial sums of the strips streamed in pieces
size_t trsm_strips_scratch(int n)
{
    return ((size_t)3 * n * MS_YLD + (size_t)(n / LEAF) * MSLOTS * MFRAG + piece_partials(n / LEAF, piece_cap(n / LEAF)) * XPART) * sizeof(double) +
           (size_t)2 * (n / LEAF + 1) * sizeof(int);
}

// B (n x nrhs, column-major, device) := L^-T L^-1 B, 64 columns per pass through the images; `state`: TRSM_STATE_INTS ints of
// device scratch; `scratch`: trsm_strips_scratch(n) bytes.
int potrs_strips(int n, const double *L, size_t ldl, const double *inv, double *B, size_t ldb, int nrhs, int *state,
                 double *scratch, hipStream_t st)
{
    if (n <= 0 || nrhs <= 0) return 0;
    if (!trsm_strips_ok(n, L, ldl)) { set_error("potrs_strips: shape not supported"); return SGPR_E_ARG; }
    const int T = n / LEAF;
    // chain class: four quarters of the strips next to the frontier, each MS_F + 1 products long
    static const int nchain_env = (int)tune("trsm_chain", 0);
    int nchain = nchain_env > 0 ? nchain_env : 4 * (MS_F + 2);
    if (nchain > 4 * T) nchain = 4 * T;
    const int piece = piece_cap(T);
    const size_t ntick = piece_tickets(T, piece);
    const int nstream = ntick < (size_t)(256 - nchain) ? (int)ntick : 256 - nchain;      // one workgroup per CU: the grid is persistent
    const size_t img = (size_t)n * MS_YLD;
    double *I0 = scratch, *I1 = scratch + img, *S = scratch + 2 * img, *M = scratch + 3 * img, *X = M + (size_t)T * MSLOTS * MFRAG;
    const size_t xbytes = piece_partials(T, piece) * XPART * sizeof(double);
    int *mflag = reinterpret_cast<int *>(X + piece_partials(T, piece) * XPART);     // T + 1 ints per triangular solve
    // The give-up words (state[2], state[6]) are cleared ONCE per call: a pass that gave up stays on record while the later
    // 64-column passes reset their tickets and ready words only (and leave at their first look at it), so the status the
    // caller reads after the last pass covers every pass.  Tunable "trsm_force_giveup_pass" (tests, through the probe
    // library): raise the forward give-up word in front of that pass.
    SGPR_HIP(hipMemsetAsync(state, 0, TRSM_STATE_INTS * sizeof(int), st));
    const int force_pass = (int)tune("trsm_force_giveup_pass", -1);
    for (int c0 = 0; c0 < nrhs; c0 += MS_NC) {
        const int nc = nrhs - c0 < MS_NC ? nrhs - c0 : MS_NC;
        hipLaunchKernelGGL(pack_rhs_kernel, dim3((n + 63) / 64), dim3(256), 0, st, n, nc, B + (size_t)c0 * ldb, ldb, I0);
        SGPR_CHECK_LAUNCH();
        hipLaunchKernelGGL(state_reset_kernel, dim3(1), dim3(64), 0, st, state, force_pass == c0 / MS_NC ? 1 : 0);
        SGPR_CHECK_LAUNCH();
        SGPR_HIP(hipMemsetAsync(I1, 0xFF, 2 * img * sizeof(double), st));       // P and S of the forward solve
        if (xbytes) SGPR_HIP(hipMemsetAsync(X, 0xFF, xbytes, st));
        SGPR_HIP(hipMemsetAsync(mflag, 0, (size_t)2 * (T + 1) * sizeof(int), st));
        TrsmArgs a{T, nchain, L, ldl, inv, I0, I1, S, M, mflag, X, piece, nc, state, state + 8, nullptr, TRSM_DBG ? (int)tune("trsm_fake_b", 0) : 0};
        const bool dbg = TRSM_DBG && getenv("SGPR_TRSM_DBG") != nullptr;
        if (dbg) { (void)hipMalloc((void **)&a.dbg, sizeof(unsigned long long) * DBGW * T); (void)hipMemset(a.dbg, 0, sizeof(unsigned long long) * DBGW * T); }
        hipLaunchKernelGGL(trsm_strips_kernel<true>, dim3(nchain + nstream), dim3(MS_T), 0, st, a);
        SGPR_CHECK_LAUNCH();
        if (dbg) dbg_report(a, "forward", st);
        // backward: in = the forward solution, out = the first image
        SGPR_HIP(hipMemsetAsync(I0, 0xFF, img * sizeof(double), st));
        SGPR_HIP(hipMemsetAsync(S, 0xFF, img * sizeof(double), st));
        if (xbytes) SGPR_HIP(hipMemsetAsync(X, 0xFF, xbytes, st));
        a.Bin = I1; a.P = I0;
        a.state = state + 4; a.ready = state + 12; a.mflag = mflag + T + 1;
        hipLaunchKernelGGL(trsm_strips_kernel<false>, dim3(nchain + nstream), dim3(MS_T), 0, st, a);
        SGPR_CHECK_LAUNCH();
        if (dbg) { dbg_report(a, "backward", st); (void)hipFree(a.dbg); }
        hipLaunchKernelGGL(unpack_rhs_kernel, dim3((n + 63) / 64), dim3(256), 0, st, n, nc, I0, B + (size_t)c0 * ldb, ldb);
        SGPR_CHECK_LAUNCH();
    }
    return 0;
}

}  // namespace sgpr

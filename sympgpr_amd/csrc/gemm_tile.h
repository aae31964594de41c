// gemm_tile.h -- device-side pieces of the fp64 MFMA product shared by gemm_f64.hip (the grid-wide
// kernel) and chol.hip (the persistent panel kernel, whose workgroups run 128 x 128 x 128 products
// as tasks): argument block, workgroup -> tile map, operand staging, the two k-loop bodies.
// See gemm_f64.hip for the design notes.
#pragma once
#include <type_traits>

#include "common.h"

namespace sgpr {
namespace tile {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// LDS stages of the 256x128 LDS-DMA body.  3 (156 KiB, tile t+2 in flight) was measured against 2 on
// one box: 8192^3 probe 73.2 vs 73.4 TFLOP/s, n = 131072 factorisation 11.17 vs 11.02 s -- the
// pipeline is not waiting for memory, so the deeper prefetch only adds outstanding traffic.
#ifndef SGPR_GEMM_STAGES
#define SGPR_GEMM_STAGES 2
#endif
constexpr int BK = 16;
constexpr int PAD = 16;

struct GemmArgs {
    int m, n, k;
    double alpha, beta;
    const double *A;
    size_t lda;
    const double *B;
    size_t ldb;
    double *C;
    size_t ldc;
    int lower;
    long diag_off;
    int transb;                  // B is (k x n) column-major ("NN" product) instead of (n x k)
    unsigned long long *stamps;  // diagnostic: per-workgroup shader-clock / real-time stamps, or null
    // tile -> workgroup map (see tile_of): super-tiles of SR x SC tiles, one per XCD at a time
    int tiles_m, tiles_n, n_sr, n_sc, n_super, tri;
    int n_full, n_grp;           // tri: full super-tiles (enumerated first), groups of 4 diagonal ones
    // lower-mode skip test in block-cyclic form: a tile is needed iff
    //   (last_row / lblk) * lpr + lpi >= (first_col / lblk) * lpc + lpj
    // single GPU: lblk = 1, lpr = lpc = 1, lpi = diag_off, lpj = 0  (row + diag_off >= col)
    int lblk, lpr, lpi, lpc, lpj;
    int dbg;  // probe switches: 8 = force the 128x128 tile shape, 16 = force the register-staged body
    unsigned long long *kdone;  // diagnostic (task-queue trace): 100 MHz time at which wave 0 left the k-loop, or null
};

// Workgroup -> tile map.  The dispatcher deals consecutive workgroup ids round-robin over the 8
// XCDs, each with a private 4 MiB L2.  Ids that land on one XCD (id % 8 equal) are handed a
// compact SR x SC block of tiles ("super-tile": 8 x 4 tiles = 2048 x 512 of C for the big
// kernel), so the 32 workgroups resident on an XCD stream only 8 A-panels + 4 B-panels through
// its L2 instead of 32 + 32 -- the operand traffic that reaches HBM drops ~5x.  This is a
// speed-only assumption: any other placement computes the same tiles.
// `tri`: square SYRK with the diagonal at 0 -- only super-tiles touching the lower triangle are
// enumerated (column-major over super-columns), so no workgroup slot is spent on an early exit.
// The super-tile is SR x SC tiles with SR*BM == 4*SC*BN (2048 x 512 of C for the 256x128 shape,
// 1024 x 256 for the 128x128 shape), which is what the triangular closed form assumes; any square
// size works (checked exhaustively on the host against the set of needed tiles).
constexpr int SR = 8;
template <int SC>
__device__ __forceinline__ bool tile_of(const GemmArgs &g, int &tile_r, int &tile_c)
{
    const int L = blockIdx.x;
    const int xcd = L & 7, j = L >> 3;
    const int S = (j / (SR * SC)) * 8 + xcd;
    if (S >= g.n_super) return false;
    const int w = j % (SR * SC);
    int sr, sc;
    if (!g.tri) {
        sc = S / g.n_sr;
        sr = S - sc * g.n_sr;
        // lower mode without the closed form (block-cyclic test, diag_off != 0): skew the rows by the
        // column so that one XCD (S % 8) is not handed the same super-row -- the empty top or the
        // full bottom of the triangle -- in every super-column
        if (g.lower) sr = (sr + sc) % g.n_sr;
    } else if (S < g.n_full) {
        // Super-column sc needs super-rows >= a = sc / 4 (4 super-columns per super-row of C).  The
        // super-tiles strictly below that first row are full; they come first, column-major:
        // columns 4a..4a+3 hold N1 - a of them each, N1 = n_sr - 1, so
        // cum(a) = 4 (a N1 - a(a-1)/2); find the largest a with cum(a) <= S.
        const int N1 = g.n_sr - 1;
        const double nsr = (double)N1;
        int a = (int)((2.0 * nsr + 1.0 - sqrt((2.0 * nsr + 1.0) * (2.0 * nsr + 1.0) - 2.0 * (double)S)) * 0.5);
        if (a < 0) a = 0;
        while (a > 0 && 4 * (a * N1 - a * (a - 1) / 2) > S) --a;
        while (4 * ((a + 1) * N1 - (a + 1) * a / 2) <= S) ++a;
        const int rem = S - 4 * (a * N1 - a * (a - 1) / 2);
        const int per = N1 - a;
        const int b = rem / per;
        sc = 4 * a + b;
        sr = a + 1 + (rem - b * per);
    } else {
        // ... then the partially filled super-tiles on the diagonal, ordered by their position b in
        // the group of four (equal fill) so that S % 8 deals every XCD the same mix.  With them
        // interleaved in column order two XCDs got all of them: 1 % (n = 65536) to 4 % (16384)
        // less work than the others, i.e. the kernel ran that much longer than its average XCD.
        const int d = S - g.n_full;
        const int b = d / g.n_grp, a = d - b * g.n_grp;
        sc = 4 * a + b;
        sr = a;
    }
    tile_r = sr * SR + (w % SR);
    tile_c = sc * SC + (w / SR);
    return tile_r < g.tiles_m && tile_c < g.tiles_n;
}

// Load one (BR x BK) operand tile: element (r, kc) = P[row0 + r + (k0 + kc) * ld], two rows
// per thread per pass.  FAST: whole tile in range and 16-B aligned -> dwordx4 loads.
template <int BR, int THREADS, bool FAST>
__device__ __forceinline__ void load_tile(const double *P, size_t ld, int row0,
                                          int k0, int rows, int kmax, int tid,
                                          double2_t (&reg)[BR * BK / (2 * THREADS)])
{
    constexpr int PASSES = BR * BK / (2 * THREADS);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int e = p * THREADS + tid;
        const int r = 2 * (e % (BR / 2));
        const int kc = e / (BR / 2);
        if constexpr (FAST) {
            const double *src = P + (size_t)(row0 + r) + (size_t)(k0 + kc) * ld;
            reg[p] = *reinterpret_cast<const double2_t *>(src);
        } else {
            // branch-free edge path: clamp the address into the matrix, zero by select (a
            // branch per element makes hipcc drain vmcnt(0) between loads)
            const int kk = min(k0 + kc, kmax - 1);
            const int r0 = min(row0 + r, rows - 1), r1 = min(row0 + r + 1, rows - 1);
            const double *col = P + (size_t)kk * ld;
            const double a = col[r0], b = col[r1];
            const bool kok = (k0 + kc) < kmax;
            double2_t v;
            v.x = (kok && (row0 + r) < rows) ? a : 0.0;
            v.y = (kok && (row0 + r + 1) < rows) ? b : 0.0;
            reg[p] = v;
        }
    }
}

template <int BR, int THREADS>
__device__ __forceinline__ void store_tile(double *S, int tid,
                                           const double2_t (&reg)[BR * BK / (2 * THREADS)])
{
    constexpr int PASSES = BR * BK / (2 * THREADS);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int e = p * THREADS + tid;
        const int r = 2 * (e % (BR / 2));
        const int kc = e / (BR / 2);
        *reinterpret_cast<double2_t *>(S + kc * (BR + PAD) + r) = reg[p];
    }
}

// "NN" form: the n-side operand is given as B (k x n) column-major, i.e. k is the contiguous index.
// A thread loads two consecutive k of one column (16 B) and scatters them into the [k][col] image.
template <int BR, int THREADS, bool FAST>
__device__ __forceinline__ void load_tile_t(const double *P, size_t ld, int col0, int k0, int cols,
                                            int kmax, int tid, double2_t (&reg)[BR * BK / (2 * THREADS)])
{
    constexpr int PASSES = BR * BK / (2 * THREADS);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int e = p * THREADS + tid;
        const int kp = 2 * (e % (BK / 2));
        const int r = e / (BK / 2);
        if constexpr (FAST) {
            reg[p] = *reinterpret_cast<const double2_t *>(P + (size_t)(k0 + kp) + (size_t)(col0 + r) * ld);
        } else {
            const int cc = min(col0 + r, cols - 1);
            const int ka = min(k0 + kp, kmax - 1), kb = min(k0 + kp + 1, kmax - 1);
            const double *col = P + (size_t)cc * ld;
            const double a = col[ka], b = col[kb];
            const bool cok = (col0 + r) < cols;
            double2_t v;
            v.x = (cok && (k0 + kp) < kmax) ? a : 0.0;
            v.y = (cok && (k0 + kp + 1) < kmax) ? b : 0.0;
            reg[p] = v;
        }
    }
}
template <int BR, int THREADS>
__device__ __forceinline__ void store_pass_t(double *S, int tid, const double2_t &v, int p)
{
    const int e = p * THREADS + tid;
    const int kp = 2 * (e % (BK / 2));
    const int r = e / (BK / 2);
    S[kp * (BR + PAD) + r] = v.x;
    S[(kp + 1) * (BR + PAD) + r] = v.y;
}

// one staging pass (a quarter / half of a tile) -> LDS; lets the k-loop slot the writes between MFMAs
template <int BR, int THREADS>
__device__ __forceinline__ void store_pass(double *S, int tid, const double2_t &v, int p)
{
    const int e = p * THREADS + tid;
    const int r = 2 * (e % (BR / 2));
    const int kc = e / (BR / 2);
    *reinterpret_cast<double2_t *>(S + kc * (BR + PAD) + r) = v;
}

template <int BM, int BN, bool FAST, bool TRANSB = false>
__device__ __forceinline__ void gemm_body(const GemmArgs &g, double *smem, int tile_r, int tile_c)
{
    constexpr int WGM = BM / 64, WGN = BN / 64;
    constexpr int THREADS = 64 * WGM * WGN;
    constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD;
    constexpr int PA = BM * BK / (2 * THREADS), PB = BN * BK / (2 * THREADS);
    static_assert(PA >= 1 && PB >= 1, "tile too small for the thread count");
    double *const sA0 = smem;
    double *const sB0 = smem + 2 * BK * LDA_S;
    const int row0 = tile_r * BM;
    const int col0 = tile_c * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WGM, wn = wave / WGM;
    const int l15 = lane & 15, l4 = lane >> 4;


    double4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};

    double2_t ra[PA], rb[PB];
    const int T = (g.k + BK - 1) / BK;
    unsigned long long st_c0 = 0, st_r0 = 0;
    if (g.stamps) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }

    // FAST (block-uniform, decided once in the kernel): every operand tile of this workgroup
    // is in range and 16-B aligned -> straight-line dwordx4 loads, no branch inside the k-loop.
    auto fetch = [&](int t) {
        const int k0 = t * BK;
        load_tile<BM, THREADS, FAST>(g.A, g.lda, row0, k0, g.m, g.k, tid, ra);
        if constexpr (TRANSB) load_tile_t<BN, THREADS, FAST>(g.B, g.ldb, col0, k0, g.n, g.k, tid, rb);
        else                  load_tile<BN, THREADS, FAST>(g.B, g.ldb, col0, k0, g.n, g.k, tid, rb);
    };
    auto store_b = [&](double *S, int p) {
        if constexpr (TRANSB) store_pass_t<BN, THREADS>(S, tid, rb[p], p);
        else                  store_pass<BN, THREADS>(S, tid, rb[p], p);
    };

    // Software pipeline (one barrier per k-step, placed where every wave still has MFMAs queued):
    //   top of step t : global loads of tile t+1 go out (register staging, 6 x 16 B per thread)
    //   kk = 0..3     : 16 MFMAs each on fragment set kk&1 while set (kk+1)&1 is being read
    //   start of kk=2 : tile t+1 is written to the other LDS buffer (its last reader finished
    //                   before the previous step's barrier)
    //   middle of kk=3: barrier; right after it the kk=0 fragments of tile t+1 are read, so the
    //                   next step starts with its operands already in registers.
    double fa[2][4], fb[2][4];
    auto load_frags = [&](int buf, int kk, int set) {
        const double *sA = sA0 + buf * BK * LDA_S + wm * 64 + l15 + (kk * 4 + l4) * LDA_S;
        const double *sB = sB0 + buf * BK * LDB_S + wn * 64 + l15 + (kk * 4 + l4) * LDB_S;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[set][i] = sB[i * 16];  // MFMA A operand <- n side
            fb[set][i] = sA[i * 16];  // MFMA B operand <- m side
        }
    };
    auto mfma_rows = [&](int set, int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };

    if (T > 0) {
        fetch(0);
        store_tile<BM, THREADS>(sA0, tid, ra);
#pragma unroll
        for (int p = 0; p < PB; ++p) store_b(sB0, p);
        __syncthreads();
        load_frags(0, 0, 0);
    }
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < T;
        if (more) fetch(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        // kk = 0
        load_frags(cur, 1, 1);
        __builtin_amdgcn_sched_barrier(0);  // reads go out FIRST: hipcc otherwise sinks them to 1-2 MFMAs before use
        mfma_rows(0, 0, 4);
        __builtin_amdgcn_sched_barrier(0);
        // kk = 1
        load_frags(cur, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 0, 4);
        __builtin_amdgcn_sched_barrier(0);
        // kk = 2: tile t+1 goes to the other LDS buffer, a slice of it after every 4 MFMAs
        load_frags(cur, 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        double *const nA = sA0 + (cur ^ 1) * BK * LDA_S;
        double *const nB = sB0 + (cur ^ 1) * BK * LDB_S;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mfma_rows(0, i, i + 1);
            if (more) {
#pragma unroll
                for (int p = i * PA / 4; p < (i + 1) * PA / 4; ++p) store_pass<BM, THREADS>(nA, tid, ra[p], p);
#pragma unroll
                for (int p = i * PB / 4; p < (i + 1) * PB / 4; ++p) store_b(nB, p);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // kk = 3
        mfma_rows(1, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (more) load_frags(cur ^ 1, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 2, 4);
        __builtin_amdgcn_sched_barrier(0);
    }

    if (g.stamps && tid == 0) {
        unsigned long long *o = g.stamps + 4 * ((size_t)tile_c * g.tiles_m + tile_r);
        o[0] = st_c0; o[1] = __builtin_amdgcn_s_memtime();
        o[2] = st_r0; o[3] = __builtin_amdgcn_s_memrealtime();
    }

    // epilogue: acc[i][j][r] is C(m = row0 + wm*64 + j*16 + l15, n = col0 + wn*64 + i*16 + 4r + l4).
    // Old values are fetched 16 at a time before any store (see gemm_body_dma: per-element
    // read-modify-write compiles to 64 dependent round trips).
    const double alpha = g.alpha, beta = g.beta;
    const bool interior = (row0 + BM <= g.m) && (col0 + BN <= g.n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double old[4][4];
        if (beta != 0.0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = col0 + wn * 64 + i * 16 + 4 * r + l4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = row0 + wm * 64 + j * 16 + l15;
                    old[r][j] = (interior || (m < g.m && n < g.n)) ? g.C[(size_t)m + (size_t)n * g.ldc] : 0.0;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = col0 + wn * 64 + i * 16 + 4 * r + l4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = row0 + wm * 64 + j * 16 + l15;
                if (interior || (m < g.m && n < g.n)) {
                    const double v = alpha * acc[i][j][r];
                    g.C[(size_t)m + (size_t)n * g.ldc] = (beta == 0.0) ? v : __builtin_fma(beta, old[r][j], v);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- fast path: LDS-DMA staging + counted LDS waits -------------------------------------------
// Interior workgroups (full, 16-B aligned tiles, k % 16 == 0) stage their operands with
// global_load_lds_dwordx4: one wave-instruction copies 1 KiB = 128 consecutive rows of one k-column
// straight into the [k][row] LDS image (lane-linear, so the 16-double row pad stays legal), no
// staging VGPRs, no ds_write, and the copy of tile t+1 is in flight under all 64 MFMAs of tile t.
// Fragment reads are explicit ds_read_b64 (hipcc fuses neighbouring reads into ds_read2_b64 at
// half the LDS rate) with counted lgkmcnt waits: 8 reads of the NEXT k-block stay in flight
// while the current block's MFMAs issue.
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
// fragments of k-block KK: fa <- n side (MFMA A operand), fb <- m side (MFMA B operand)
template <int KK, int LDA_S, int LDB_S>
__device__ __forceinline__ void read_frags(unsigned aA, unsigned aB, double (&fa)[4], double (&fb)[4])
{
    fa[0] = ds_read_f64<(KK * 4 * LDB_S + 0) * 8>(aB);
    fb[0] = ds_read_f64<(KK * 4 * LDA_S + 0) * 8>(aA);
    fa[1] = ds_read_f64<(KK * 4 * LDB_S + 16) * 8>(aB);
    fb[1] = ds_read_f64<(KK * 4 * LDA_S + 16) * 8>(aA);
    fa[2] = ds_read_f64<(KK * 4 * LDB_S + 32) * 8>(aB);
    fb[2] = ds_read_f64<(KK * 4 * LDA_S + 32) * 8>(aA);
    fa[3] = ds_read_f64<(KK * 4 * LDB_S + 48) * 8>(aB);
    fb[3] = ds_read_f64<(KK * 4 * LDA_S + 48) * 8>(aA);
}
#define SGPR_LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int BM, int BN, int STAGES>
__device__ __forceinline__ void gemm_body_dma(const GemmArgs &g, double *smem, int tile_r, int tile_c)
{
    constexpr int WGM = BM / 64, WGN = BN / 64;
    constexpr int NW = WGM * WGN;
    constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD;
    constexpr int QA = BK * (BM / 128) / NW, QB = BK * (BN / 128) / NW;  // DMA instructions per wave
    static_assert(QA >= 1 && QB >= 1 && BM % 128 == 0 && BN % 128 == 0, "tile / wave count mismatch");
    static_assert(STAGES == 2 || STAGES == 3, "two or three LDS stages");
    double *const sA0 = smem;
    double *const sB0 = smem + STAGES * BK * LDA_S;
    const int row0 = tile_r * BM, col0 = tile_c * BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int l15 = lane & 15, l4 = lane >> 4;

    double4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};

    const int T = g.k / BK;
    unsigned long long st_c0 = 0, st_r0 = 0;
    if (g.stamps) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }

    // this wave's DMA sources (per lane: 2 consecutive rows of one k-column) and LDS row starts
    const double *srcA[QA], *srcB[QB];
    int offA[QA], offB[QB];
#pragma unroll
    for (int j = 0; j < QA; ++j) {
        const int q = wave + NW * j, kc = q / (BM / 128), seg = q % (BM / 128);
        srcA[j] = g.A + (size_t)(row0 + seg * 128 + 2 * lane) + (size_t)kc * g.lda;
        offA[j] = kc * LDA_S + seg * 128;
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const int q = wave + NW * j, kc = q / (BN / 128), seg = q % (BN / 128);
        srcB[j] = g.B + (size_t)(col0 + seg * 128 + 2 * lane) + (size_t)kc * g.ldb;
        offB[j] = kc * LDB_S + seg * 128;
    }
    const size_t stepA = (size_t)BK * g.lda, stepB = (size_t)BK * g.ldb;
    // one LDS-DMA instruction: q < QA -> A piece q, else B piece q - QA
    auto dma_one = [&](int t, int buf, int q) {
        if (q < QA) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(srcA[q] + (size_t)t * stepA),
                (__attribute__((address_space(3))) void *)(sA0 + buf * BK * LDA_S + offA[q]), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(srcB[q - QA] + (size_t)t * stepB),
                (__attribute__((address_space(3))) void *)(sB0 + buf * BK * LDB_S + offB[q - QA]), 16, 0, 0);
        }
    };
    auto dma = [&](int t, int buf) {
#pragma unroll
        for (int q = 0; q < QA + QB; ++q) dma_one(t, buf, q);
    };
    constexpr int NQ = QA + QB;

    double fa0[4], fb0[4], fa1[4], fb1[4];
    auto mfma_rows = [&](const double (&fa)[4], const double (&fb)[4], int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    const unsigned baseA = lds_addr(sA0 + wm * 64 + l15 + l4 * LDA_S);
    const unsigned baseB = lds_addr(sB0 + wn * 64 + l15 + l4 * LDB_S);

    // AHEAD = how many tiles beyond the current one are in flight or landed: 1 (two stages) or 2.
    constexpr int AHEAD = STAGES - 1;
    if (T > 0) {
        dma(0, 0);
        if (STAGES == 3 && T > 1) {
            dma(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");   // tile 0 landed, tile 1 in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        read_frags<0, LDA_S, LDB_S>(baseA, baseB, fa0, fb0);
    }
    // One k-step.  ISSUE (compile time): tile t+AHEAD exists and is requested in this step; NEXT:
    // tile t+1 exists.  The steady-state body has no branch at all: a single scalar branch in this
    // instruction stream costs ~50 cycles of MFMA issue per use (measured: seven `if (more)` tests
    // per step = 350 of 8700 cycles), so the last step(s) are peeled.
    // Buffers: tile t lives in buffer t % STAGES.  The copy of tile t+AHEAD targets the buffer tile
    // t-1 occupied, which every wave released at the barrier of step t-1.
    auto kstep = [&](int t, int cur, auto issue_tag, auto next_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value, NEXT = decltype(next_tag)::value;
        const int nxt = (cur + 1 == STAGES) ? 0 : cur + 1;
        const int tgt = (STAGES == 2) ? nxt : ((nxt + 1 == STAGES) ? 0 : nxt + 1);
        const unsigned aA = baseA + cur * (BK * LDA_S * 8), aB = baseB + cur * (BK * LDB_S * 8);
        __builtin_amdgcn_sched_barrier(0);
        // kk = 0 and 1: the copy of tile t+AHEAD goes out one LDS-DMA instruction per MFMA row, so the
        // matrix pipe never waits behind a burst of address arithmetic + DMA issue
        read_frags<1, LDA_S, LDB_S>(aA, aB, fa1, fb1);
        SGPR_LGKM_WAIT(8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mfma_rows(fa0, fb0, i, i + 1);
            if constexpr (ISSUE) {
#pragma unroll
                for (int q = i * NQ / 8; q < (i + 1) * NQ / 8; ++q) dma_one(t + AHEAD, tgt, q);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        read_frags<2, LDA_S, LDB_S>(aA, aB, fa0, fb0);
        SGPR_LGKM_WAIT(8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mfma_rows(fa1, fb1, i, i + 1);
            if constexpr (ISSUE) {
#pragma unroll
                for (int q = (4 + i) * NQ / 8; q < (5 + i) * NQ / 8; ++q) dma_one(t + AHEAD, tgt, q);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // kk = 2
        read_frags<3, LDA_S, LDB_S>(aA, aB, fa1, fb1);
        SGPR_LGKM_WAIT(8);
        mfma_rows(fa0, fb0, 0, 4);
        __builtin_amdgcn_sched_barrier(0);
        // kk = 3: first half, then the step's only barrier, then the next tile's first fragments
        SGPR_LGKM_WAIT(0);
        mfma_rows(fa1, fb1, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT) {
            // this wave's share of tile t+1 has landed (with three stages the copies of tile t+2
            // issued above may still be in flight)
            if constexpr (STAGES == 3 && ISSUE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
            else                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned nA = baseA + nxt * (BK * LDA_S * 8), nB = baseB + nxt * (BK * LDB_S * 8);
            read_frags<0, LDA_S, LDB_S>(nA, nB, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_rows(fa1, fb1, 2, 4);
        __builtin_amdgcn_sched_barrier(0);
        return nxt;
    };
    {
        int t = 0, cur = 0;
        for (; t + AHEAD < T; ++t) cur = kstep(t, cur, std::true_type(), std::true_type());
        for (; t + 1 < T; ++t) cur = kstep(t, cur, std::false_type(), std::true_type());
        if (t < T) kstep(t, cur, std::false_type(), std::false_type());
    }
    SGPR_LGKM_WAIT(0);

    if (g.stamps && tid == 0) {
        unsigned long long *o = g.stamps + 4 * ((size_t)tile_c * g.tiles_m + tile_r);
        o[0] = st_c0; o[1] = __builtin_amdgcn_s_memtime();
        o[2] = st_r0; o[3] = __builtin_amdgcn_s_memrealtime();
    }
    if (g.kdone && tid == 0) *g.kdone = __builtin_amdgcn_s_memrealtime();

    // interior tile: no bounds checks.  acc[i][j][r] is C(row0 + wm*64 + j*16 + l15, col0 + wn*64 + i*16 + 4r + l4)
    // The old values of C are fetched 16 at a time BEFORE any of them is overwritten: written as
    // `*c = fma(beta, *c, v)` per element, hipcc cannot rule out aliasing between one element's store
    // and the next one's load and emits 64 dependent load -> wait -> store round trips per thread
    // (~50 us per tile, the whole fixed cost of a short-k launch: k = 512 ran at 42 TFLOP/s).
    const double alpha = g.alpha, beta = g.beta;
    double *const cbase = g.C + (size_t)(row0 + wm * 64 + l15) + (size_t)(col0 + wn * 64 + l4) * g.ldc;
    // ... and the fetch of the NEXT sixteen goes out before the current sixteen are stored.  (Round 5: the ~15 us a tile spends
    // outside its k-loop are bandwidth, not latency -- all 256 workgroups of a wave of tiles reach their epilogue together,
    // 134 MB of C at 8 TB/s; the pipelined fetch gains 0.8 us of them: profiles/r05/gemm_k_epilogue.txt.)
    double old[2][4][4];
    auto fetch = [&](int i, double (&o)[4][4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[r][j] = cbase[(size_t)(j * 16) + (size_t)(i * 16 + 4 * r) * g.ldc];
    };
    if (beta != 0.0) fetch(0, old[0]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (beta != 0.0 && i + 1 < 4) fetch(i + 1, old[(i + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        // the sixteen of this round have landed, the next sixteen may still be in flight (loads return in order)
        if (beta != 0.0) {
            if (i + 1 < 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else           asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double v = alpha * acc[i][j][r];
                cbase[(size_t)(j * 16) + (size_t)(i * 16 + 4 * r) * g.ldc] = (beta == 0.0) ? v : __builtin_fma(beta, old[i & 1][r][j], v);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}


}  // namespace tile
}  // namespace sgpr

// gemm_f64.hip -- C = beta C + alpha A B^T in fp64 on the gfx950 matrix cores
// (v_mfma_f64_16x16x4_f64).  This is where the n^3/3 flop of the Cholesky
// (scipy.linalg.cholesky -> LAPACK dpotrf at python/functions/func.py:166,184,193) are spent:
// the SYRK/GEMM trailing updates and the panel solves all run through this kernel.
//
// Roofline: fp64 MFMA.  One 16x16x4 MFMA = 2048 flop per wave for one fp64 operand register
// per side, so the kernel is arranged to keep the matrix pipe issuing back to back:
//   * workgroup tile 256 x 128 (8 waves, each 64 x 64 = 16 accumulators of 4 fp64 = 128
//     VGPRs -> two waves per SIMD fit), k-step 16, LDS double-buffered, one barrier per k-step,
//     the next k-tile's global loads in flight under the current tile's 64 MFMAs per wave;
//   * both operands are "row index contiguous, k strided" (column-major panels of the
//     factor), which is exactly the MFMA A/B fragment order (lane&15 = row, lane>>4 = k), so the
//     LDS image is the global image: [k][row] with the row stride padded by 16 doubles
//     (2*stride mod 64 banks = 32 -> the two k-rows of a 32-lane ds_read_b64 group hit
//     disjoint bank halves);
//   * the MFMA is fed with the B(n)-side as its A operand and the A(m)-side as its B operand:
//     the accumulator then has lane&15 = m (memory-contiguous in column-major C) and
//     4*reg+(lane>>4) = n, so every C access of a 16-lane quarter is one full 128-B line.
//     (fp64 C/D layout: col = lane&15, row = (lane>>4) + 4*reg -- NOT the f32 map.)
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"

namespace sgpr {

namespace {

// optional per-launch HIP-event timing (bench.py's roofline leg); off by default
struct ProfRec { hipEvent_t a, b; double flop; int big; int m, n, k, lower; };
struct Prof { bool on = false; std::vector<ProfRec> recs; };
Prof g_prof;
int g_dbg = 0;
unsigned long long *g_stamps = nullptr;  // set by gemm_set_stamps (diagnostics only)

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

    // interior tile: no bounds checks.  acc[i][j][r] is C(row0 + wm*64 + j*16 + l15, col0 + wn*64 + i*16 + 4r + l4)
    // The old values of C are fetched 16 at a time BEFORE any of them is overwritten: written as
    // `*c = fma(beta, *c, v)` per element, hipcc cannot rule out aliasing between one element's store
    // and the next one's load and emits 64 dependent load -> wait -> store round trips per thread
    // (~50 us per tile, the whole fixed cost of a short-k launch: k = 512 ran at 42 TFLOP/s).
    const double alpha = g.alpha, beta = g.beta;
    double *const cbase = g.C + (size_t)(row0 + wm * 64 + l15) + (size_t)(col0 + wn * 64 + l4) * g.ldc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double old[4][4];
        if (beta != 0.0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) old[r][j] = cbase[(size_t)(j * 16) + (size_t)(i * 16 + 4 * r) * g.ldc];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double v = alpha * acc[i][j][r];
                cbase[(size_t)(j * 16) + (size_t)(i * 16 + 4 * r) * g.ldc] = (beta == 0.0) ? v : __builtin_fma(beta, old[r][j], v);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// second launch-bound argument = waves per SIMD: both tile shapes are sized for TWO waves per SIMD
// (<= 256 VGPRs): one 512-thread workgroup per CU (256x128), or two independent 256-thread
// workgroups per CU (128x128) whose barriers do not line up.
template <int BM, int BN>
__global__ __launch_bounds__(64 * (BM / 64) * (BN / 64), (BM >= 128 ? 2 : 1)) void gemm_nt_kernel(const GemmArgs g)
{
    constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD;
    // SGPR_GEMM_STAGES for the one-workgroup-per-CU shape (3 = 156 KiB of the CU's 160), two otherwise
    constexpr int STAGES = (BM == 256 && BN == 128) ? SGPR_GEMM_STAGES : 2;
    __shared__ double smem[STAGES * BK * (LDA_S + LDB_S)];
    int tile_r, tile_c;
    if (!tile_of<(SR * BM) / (4 * BN)>(g, tile_r, tile_c)) return;
    const int row0 = tile_r * BM, col0 = tile_c * BN;
    if (g.lower) {
        const long rb = ((long)min(row0 + BM, g.m) - 1) / g.lblk, cb = (long)col0 / g.lblk;
        if (rb * g.lpr + g.lpi < cb * g.lpc + g.lpj) return;
    }
    const bool aligned = ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0) && (((g.lda | g.ldb) & 1) == 0);
    const bool fast = aligned && (row0 + BM <= g.m) && (col0 + BN <= g.n) && (g.k % BK == 0);
    if (g.transb) {
        if (fast) gemm_body<BM, BN, true, true>(g, smem, tile_r, tile_c);
        else      gemm_body<BM, BN, false, true>(g, smem, tile_r, tile_c);
        return;
    }
    if constexpr (BM % 128 == 0) {
        if (fast && !(g.dbg & 16)) { gemm_body_dma<BM, BN, STAGES>(g, smem, tile_r, tile_c); return; }
    }
    if (fast) gemm_body<BM, BN, true>(g, smem, tile_r, tile_c);
    else      gemm_body<BM, BN, false>(g, smem, tile_r, tile_c);
}

}  // namespace

int gemm_nt(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
            size_t ldb, double beta, double *C, size_t ldc, int lower, long diag_off,
            hipStream_t st)
{
    const int bc[5] = {1, 1, (int)diag_off, 1, 0};
    return gemm_nt_bc(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lower, bc, st);
}

static int gemm_launch(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
                       size_t ldb, double beta, double *C, size_t ldc, int lower, const int *bc, int transb,
                       hipStream_t st);

// `bc` = {blk, pr, pi, pc, pj}: the block-cyclic form of the lower-mode skip test (GemmArgs)
int gemm_nt_bc(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
               size_t ldb, double beta, double *C, size_t ldc, int lower, const int *bc,
               hipStream_t st)
{
    return gemm_launch(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lower, bc, 0, st);
}

static int gemm_launch(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
                       size_t ldb, double beta, double *C, size_t ldc, int lower, const int *bc, int transb,
                       hipStream_t st)
{
    const bool plain = bc[0] == 1 && bc[1] == 1 && bc[3] == 1 && bc[4] == 0;
    const long diag_off = plain ? bc[2] : 1;  // != 0 disables the triangular tile enumeration
    if (bc[0] < 1 || bc[1] < 1 || bc[3] < 1) { set_error("gemm_nt: bad block-cyclic descriptor"); return SGPR_E_ARG; }
    if (m < 0 || n < 0 || k < 0) { set_error("gemm_nt: negative extent"); return SGPR_E_ARG; }
    if (m == 0 || n == 0) return 0;
    if (lda < (size_t)m || ldb < (size_t)(transb ? k : n) || ldc < (size_t)m) {
        set_error("gemm: leading dimension too small");
        return SGPR_E_ARG;
    }
    GemmArgs g{m, n, k, alpha, beta, A, lda, B, ldb, C, ldc, lower, diag_off, transb, g_stamps, 0, 0, 0, 0, 0, 0,
               0, 0, bc[0], bc[1], bc[2], bc[3], bc[4], g_dbg};
    auto set_map = [&](int bm, int bn) {
        g.tiles_m = (m + bm - 1) / bm;
        g.tiles_n = (n + bn - 1) / bn;
        const int SC = (SR * bm) / (4 * bn);  // 4 for 256x128, 2 for 128x128
        g.n_sr = (g.tiles_m + SR - 1) / SR;
        g.n_sc = (g.tiles_n + SC - 1) / SC;
        g.tri = (lower && diag_off == 0 && m == n) ? 1 : 0;
        if (g.tri) {
            // full super-tiles below the first needed row of every super-column, then 4 diagonal
            // slots per group (slots of a partial last group that do not exist exit at once)
            long full = 0;
            for (int sc = 0; sc < g.n_sc; ++sc) full += std::max(g.n_sr - sc / 4 - 1, 0);
            g.n_full = (int)full;
            g.n_grp = (g.n_sc + 3) / 4;
            g.n_super = g.n_full + 4 * g.n_grp;
        }
        if (!g.tri) g.n_super = g.n_sr * g.n_sc;
        return (unsigned)(((g.n_super + 7) / 8) * 8 * SR * SC);
    };
    ProfRec rec{};
    if (g_prof.on) {
        // algorithmic flop of this launch: 2k per updated element (lower: on/below the diagonal)
        double elems = (double)m * n;
        if (lower && plain) {
            elems = 0.0;
            for (int j = 0; j < n; ++j) {
                long first = (long)j - diag_off;  // first row with row + diag_off >= col
                if (first < 0) first = 0;
                if (first < m) elems += (double)(m - first);
            }
        }
        rec.flop = 2.0 * k * elems;
        rec.m = m; rec.n = n; rec.k = k; rec.lower = lower;
        SGPR_HIP(hipEventCreate(&rec.a));
        SGPR_HIP(hipEventCreate(&rec.b));
        SGPR_HIP(hipEventRecord(rec.a, st));
    }
    // big tile once it yields enough workgroups to fill 256 CUs, small tile below that
    const long big = (long)((m + 255) / 256) * ((n + 127) / 128);
    // Tile choice, measured on MI355X: at 8192^3 two independent 128x128 workgroups per CU reach
    // 71.5 TFLOP/s against 69.5 for one 256x128 workgroup (their barriers do not line up), but
    // inside the n = 131072 factorisation the 256x128 shape wins (61.7 vs 58.4 TFLOP/s overall:
    // a third less operand traffic per flop, and the triangular tile map below needs its 2:1
    // aspect).  So: 256x128 whenever it yields >= 256 workgroups, 128x128 below.
    // SGPR_GEMM_TILE=small forces the 128x128 shape (A/B experiments).
    static const bool prefer_big = [] { const char *e = getenv("SGPR_GEMM_TILE"); return !(e && e[0] == 's'); }();
    if (n <= 128 && m <= 32768 && !(g_dbg & 8)) {
        // one column tile (the in-place panel solve against an inverted leaf, k = n <= 128): a
        // latency problem, not a throughput one -- 64-row tiles quadruple the workgroup count and
        // halve the per-workgroup critical path (two waves, register-staged operands)
        const dim3 grid(set_map(64, 128));
        hipLaunchKernelGGL((gemm_nt_kernel<64, 128>), grid, dim3(128), 0, st, g);
    } else if (prefer_big && !(g_dbg & 8) && (big >= 256 || (m >= 256 && n == 128))) {
        const dim3 grid(set_map(256, 128));
        hipLaunchKernelGGL((gemm_nt_kernel<256, 128>), grid, dim3(512), 0, st, g);
        rec.big = 1;
    } else {
        const dim3 grid(set_map(128, 128));
        hipLaunchKernelGGL((gemm_nt_kernel<128, 128>), grid, dim3(256), 0, st, g);
    }
    SGPR_CHECK_LAUNCH();
    if (g_prof.on) {
        SGPR_HIP(hipEventRecord(rec.b, st));
        g_prof.recs.push_back(rec);
    }
    return 0;
}

// C (m x n) = beta C + alpha A (m x k) B (k x n): the "NN" product (B's k index contiguous)
int gemm_nn(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B, size_t ldb,
            double beta, double *C, size_t ldc, hipStream_t st)
{
    const int bc[5] = {1, 1, 0, 1, 0};
    return gemm_launch(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 0, bc, 1, st);
}

void gemm_set_stamps(unsigned long long *dev_buf) { g_stamps = dev_buf; }
void gemm_set_debug(int bits) { g_dbg = bits; }

void gemm_profile_begin()
{
    for (auto &r : g_prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.recs.clear();
    g_prof.on = true;
}

// out[0..2]: big-tile launches / flop / ms; out[3..5]: small-tile; out[6..7]: largest launch flop / ms
// per-launch records of the last profile window: 6 doubles each (m, n, k, lower, big, ms)
static std::vector<double> g_last_launches;
int gemm_profile_launches(double *buf, int max_records)
{
    const int n = (int)(g_last_launches.size() / 6);
    if (buf)
        for (int i = 0; i < n && i < max_records; ++i)
            for (int j = 0; j < 6; ++j) buf[6 * i + j] = g_last_launches[6 * i + j];
    return n;
}

int gemm_profile_end(double *out)
{
    g_prof.on = false;
    g_last_launches.clear();
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    for (auto &r : g_prof.recs) {
        SGPR_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        SGPR_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        const int o = r.big ? 0 : 3;
        out[o] += 1.0; out[o + 1] += r.flop; out[o + 2] += ms;
        if (r.flop > out[6]) { out[6] = r.flop; out[7] = ms; }
        for (double v : {(double)r.m, (double)r.n, (double)r.k, (double)r.lower, (double)r.big, (double)ms})
            g_last_launches.push_back(v);
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.recs.clear();
    return 0;
}

}  // namespace sgpr

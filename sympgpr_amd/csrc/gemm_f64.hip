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
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"
#include "gemm_tile.h"

namespace sgpr {

namespace {

// Optional per-launch HIP-event timing (bench.py's roofline pass); off by default.  The events come
// from a fixed pool created on the first gemm_profile_begin() and reused by every later window: the
// launch path never creates or destroys an event (round 1 created two per launch -- thousands of
// live events per step).  All state is behind one mutex: several fit handles may launch from their
// own threads while a window is open.
struct ProfRec { double flop; int big; int m, n, k, lower, overlap; };
struct Prof {
    std::mutex mu;
    std::atomic<bool> on{false};       // read outside the mutex on the launch path
    std::vector<hipEvent_t> pool;      // 2 * PROF_POOL events, created once per process
    std::vector<ProfRec> recs;         // record i uses pool[2 i], pool[2 i + 1]
    long dropped = 0;                  // launches beyond the pool (not timed)
};
constexpr int PROF_POOL = 8192;
Prof g_prof;
thread_local int t_overlap = 0;        // set by the look-ahead driver: launches share the device

using namespace tile;

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
                       hipStream_t st, unsigned long long *stamps = nullptr, int dbg = 0);

// `bc` = {blk, pr, pi, pc, pj}: the block-cyclic form of the lower-mode skip test (GemmArgs)
int gemm_nt_bc(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
               size_t ldb, double beta, double *C, size_t ldc, int lower, const int *bc,
               hipStream_t st)
{
    // Products deeper than 8192 run as back-to-back launches of <= 8192 columns each (C re-read per launch, negligible
    // beside the flop), so that the 32 workgroups sharing a super-tile's operand panels restart in step instead of drifting
    // apart over a 65536-deep loop.  Measured on one MI355X (profiles/r03/kmax.md): m = 65536 lower, k = 65536: 71.6 -> 72.6
    // TFLOP/s (16384: 72.0), k = 32768: 71.5 -> 72.6; the n = 131072 step 70.0 -> 71.1.  SGPR_GEMM_KMAX=<k> overrides, 0 = off.
    static const int kmax = [] { const char *e = getenv("SGPR_GEMM_KMAX"); return e ? atoi(e) : 8192; }();
    if (kmax >= 128 && k > kmax) {
        const int nchunk = (k + kmax - 1) / kmax;
        const int step = ((k + nchunk - 1) / nchunk + 127) / 128 * 128;
        for (int k0 = 0; k0 < k; k0 += step) {
            const int rc = gemm_launch(m, n, std::min(step, k - k0), alpha, A + (size_t)k0 * lda, lda, B + (size_t)k0 * ldb, ldb,
                                       k0 == 0 ? beta : 1.0, C, ldc, lower, bc, 0, st);
            if (rc) return rc;
        }
        return 0;
    }
    return gemm_launch(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lower, bc, 0, st);
}

static int gemm_launch(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
                       size_t ldb, double beta, double *C, size_t ldc, int lower, const int *bc, int transb,
                       hipStream_t st, unsigned long long *stamps, int dbg)
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
    GemmArgs g{m, n, k, alpha, beta, A, lda, B, ldb, C, ldc, lower, diag_off, transb, stamps, 0, 0, 0, 0, 0, 0,
               0, 0, bc[0], bc[1], bc[2], bc[3], bc[4], dbg};
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
    // profile window open: take the next event pair of the pool (none left: the launch goes untimed)
    int slot = -1;
    ProfRec rec{};
    if (g_prof.on.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_prof.mu);
        if (g_prof.on.load(std::memory_order_relaxed)) {
            if ((int)g_prof.recs.size() < PROF_POOL) {
                // algorithmic flop of this launch: 2k per updated element (lower: on/below the diagonal)
                double elems = (double)m * n;
                if (lower && plain) {
                    elems = 0.0;
                    for (int j = 0; j < n; ++j) {
                        long first = (long)j - diag_off;  // first row with row + diag_off >= col
                        if (first < 0) first = 0;
                        if (first < m) elems += (double)(m - first);
                    }
                } else if (lower) {
                    // local piece of a block-cyclic matrix: element (i, j) counts iff its GLOBAL position is
                    // on or below the diagonal: block (i / blk) pr + pi against (j / blk) pc + pj, then the
                    // offsets inside the block
                    elems = 0.0;
                    const long blk = bc[0];
                    for (long j = 0; j < n; ++j) {
                        const long cbg = (j / blk) * bc[3] + bc[4], jo = j % blk;
                        for (long rb = 0; rb * blk < m; ++rb) {
                            const long rbg = rb * bc[1] + bc[2], rows = std::min(blk, (long)m - rb * blk);
                            if (rbg > cbg) elems += (double)rows;
                            else if (rbg == cbg) elems += (double)std::max(0L, rows - jo);
                        }
                    }
                }
                rec.flop = 2.0 * k * elems;
                rec.m = m; rec.n = n; rec.k = k; rec.lower = lower; rec.overlap = t_overlap;
                slot = (int)g_prof.recs.size();
                g_prof.recs.push_back(rec);
            } else {
                ++g_prof.dropped;
            }
        }
    }
    if (slot >= 0) SGPR_HIP(hipEventRecord(g_prof.pool[2 * slot], st));
    // big tile once it yields enough workgroups to fill 256 CUs, small tile below that
    const long big = (long)((m + 255) / 256) * ((n + 127) / 128);
    // Tile choice, measured on MI355X: at 8192^3 two independent 128x128 workgroups per CU reach
    // 71.5 TFLOP/s against 69.5 for one 256x128 workgroup (their barriers do not line up), but
    // inside the n = 131072 factorisation the 256x128 shape wins (61.7 vs 58.4 TFLOP/s overall:
    // a third less operand traffic per flop, and the triangular tile map below needs its 2:1
    // aspect).  So: 256x128 whenever it yields >= 256 workgroups, 128x128 below.
    // tunable "gemm_small_tile" = 1 forces the 128x128 shape (A/B experiments, sgpr_probe_tune).
    static const bool prefer_big = tune("gemm_small_tile", 0) == 0;
    // tunable "gemm_small_mb" = <MB>: take the 128x128 shape for products whose operand panels are smaller than
    // that (experiments; default off).  Alone, lower-triangular m = 15360, 256x128 vs 128x128 tiles:
    // k = 256 47.4 vs 55.8, 512 49.0 vs 52.1, 1024 56.8 vs 61.2, 2048 61.2 vs 64.3 TFLOP/s -- but inside the
    // blocked factorisation, beside the panel stream, the small shape LOSES (n = 16384 39.0 vs 34.5 ms,
    // n = 32768 231 vs 208 ms): two of its workgroups share a CU and its LDS bandwidth with nothing to spare.
    static const double small_mb = tune("gemm_small_mb", 0.0);
    const double op_bytes = 8.0 * (double)k * ((A == B && lda == ldb) ? (double)std::max(m, n) : (double)m + n);
    const bool small_k = op_bytes <= small_mb * 1e6 && m > 128 && n > 128;
    if (n <= 128 && m <= 32768 && !(dbg & 8)) {
        // one column tile (the in-place panel solve against an inverted leaf, k = n <= 128): a
        // latency problem, not a throughput one -- 64-row tiles quadruple the workgroup count and
        // halve the per-workgroup critical path (two waves, register-staged operands)
        const dim3 grid(set_map(64, 128));
        hipLaunchKernelGGL((gemm_nt_kernel<64, 128>), grid, dim3(128), 0, st, g);
    } else if (prefer_big && !(dbg & 8) && !small_k && (big >= 256 || (m >= 256 && n == 128))) {
        const dim3 grid(set_map(256, 128));
        hipLaunchKernelGGL((gemm_nt_kernel<256, 128>), grid, dim3(512), 0, st, g);
        rec.big = 1;
    } else {
        const dim3 grid(set_map(128, 128));
        hipLaunchKernelGGL((gemm_nt_kernel<128, 128>), grid, dim3(256), 0, st, g);
    }
    SGPR_CHECK_LAUNCH();
    if (slot >= 0) {
        SGPR_HIP(hipEventRecord(g_prof.pool[2 * slot + 1], st));
        if (rec.big) {
            std::lock_guard<std::mutex> lock(g_prof.mu);
            if (slot < (int)g_prof.recs.size()) g_prof.recs[slot].big = 1;
        }
    }
    return 0;
}

// diagnostics only (libsympgpr_probe.so): one product with per-workgroup clock stamps and the
// tile-shape / staging switches (8 = 128x128 tiles, 16 = register-staged body); nothing global
int gemm_nt_diag(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B, size_t ldb,
                 double beta, double *C, size_t ldc, int lower, unsigned long long *stamps, int dbg, hipStream_t st)
{
    const int bc[5] = {1, 1, 0, 1, 0};
    return gemm_launch(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lower, bc, 0, st, stamps, dbg);
}

// C (m x n) = beta C + alpha A (m x k) B (k x n): the "NN" product (B's k index contiguous)
int gemm_nn(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B, size_t ldb,
            double beta, double *C, size_t ldc, hipStream_t st)
{
    const int bc[5] = {1, 1, 0, 1, 0};
    return gemm_launch(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 0, bc, 1, st);
}

void gemm_set_overlap(int on) { t_overlap = on; }

int gemm_profile_begin()
{
    std::lock_guard<std::mutex> lock(g_prof.mu);
    if (g_prof.pool.empty()) {
        g_prof.pool.resize(2 * (size_t)PROF_POOL);
        for (size_t i = 0; i < g_prof.pool.size(); ++i) {
            const hipError_t e = hipEventCreate(&g_prof.pool[i]);
            if (e != hipSuccess) {
                for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(g_prof.pool[j]);
                g_prof.pool.clear();
                return hip_fail(e, "hipEventCreate (profile pool)", __FILE__, __LINE__);
            }
        }
        g_prof.recs.reserve(PROF_POOL);
    }
    g_prof.recs.clear();
    g_prof.dropped = 0;
    g_prof.on = true;
    return 0;
}

// per-launch records of the last profile window: 7 doubles each (m, n, k, lower, big, ms, overlap)
static std::vector<double> g_last_launches;
int gemm_profile_launches(double *buf, int max_records)
{
    std::lock_guard<std::mutex> lock(g_prof.mu);
    const int n = (int)(g_last_launches.size() / 7);
    if (buf)
        for (int i = 0; i < n && i < max_records; ++i)
            for (int j = 0; j < 7; ++j) buf[7 * i + j] = g_last_launches[7 * i + j];
    return n;
}

// out[0..2]: big-tile launches / flop / ms that ran alone on the device; out[3..5]: small-tile launches
// (all); out[6..7]: largest launch flop / ms; out[8..10]: big-tile launches / flop / ms issued inside the
// look-ahead driver (two streams share the chip: their durations overlap); out[11]: launches not timed
int gemm_profile_end(double *out)
{
    std::lock_guard<std::mutex> lock(g_prof.mu);
    g_prof.on = false;
    g_last_launches.clear();
    for (int i = 0; i < 12; ++i) out[i] = 0.0;
    out[11] = (double)g_prof.dropped;
    for (size_t i = 0; i < g_prof.recs.size(); ++i) {
        const ProfRec &r = g_prof.recs[i];
        SGPR_HIP(hipEventSynchronize(g_prof.pool[2 * i + 1]));
        float ms = 0.f;
        SGPR_HIP(hipEventElapsedTime(&ms, g_prof.pool[2 * i], g_prof.pool[2 * i + 1]));
        const int o = r.big ? (r.overlap ? 8 : 0) : 3;
        out[o] += 1.0; out[o + 1] += r.flop; out[o + 2] += ms;
        if (r.flop > out[6]) { out[6] = r.flop; out[7] = ms; }
        for (double v : {(double)r.m, (double)r.n, (double)r.k, (double)r.lower, (double)r.big, (double)ms, (double)r.overlap})
            g_last_launches.push_back(v);
    }
    g_prof.recs.clear();
    return 0;
}

}  // namespace sgpr

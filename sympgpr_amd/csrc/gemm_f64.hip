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
#include <vector>

#include "common.h"

namespace sgpr {

namespace {

// optional per-launch HIP-event timing (bench.py's roofline leg); off by default
struct ProfRec { hipEvent_t a, b; double flop; int big; };
struct Prof { bool on = false; std::vector<ProfRec> recs; };
Prof g_prof;

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

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
};

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
        const double *src = P + (size_t)(row0 + r) + (size_t)(k0 + kc) * ld;
        if constexpr (FAST) {
            reg[p] = *reinterpret_cast<const double2_t *>(src);
        } else {
            const bool kok = (k0 + kc) < kmax;
            double2_t v;
            v.x = (kok && (row0 + r) < rows) ? src[0] : 0.0;
            v.y = (kok && (row0 + r + 1) < rows) ? src[1] : 0.0;
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

template <int BM, int BN>
__global__ __launch_bounds__(64 * (BM / 64) * (BN / 64)) void gemm_nt_kernel(const GemmArgs g)
{
    constexpr int WGM = BM / 64, WGN = BN / 64;
    constexpr int THREADS = 64 * WGM * WGN;
    constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD;
    constexpr int PA = BM * BK / (2 * THREADS), PB = BN * BK / (2 * THREADS);
    static_assert(PA >= 1 && PB >= 1, "tile too small for the thread count");
    __shared__ double smem[2 * BK * (LDA_S + LDB_S)];
    double *const sA0 = smem;
    double *const sB0 = smem + 2 * BK * LDA_S;

    const int row0 = blockIdx.x * BM;
    const int col0 = blockIdx.y * BN;
    if (g.lower && (long)min(row0 + BM, g.m) - 1 + g.diag_off < (long)col0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WGM, wn = wave / WGM;
    const int l15 = lane & 15, l4 = lane >> 4;

    // block-uniform: can the interior k-tiles use unpredicated 16-B loads?
    const bool alignedA = (((uintptr_t)g.A & 15) == 0) && ((g.lda & 1) == 0);
    const bool alignedB = (((uintptr_t)g.B & 15) == 0) && ((g.ldb & 1) == 0);
    const bool fullA = alignedA && (row0 + BM <= g.m);
    const bool fullB = alignedB && (col0 + BN <= g.n);

    double4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};

    double2_t ra[PA], rb[PB];
    const int T = (g.k + BK - 1) / BK;

    auto fetch = [&](int t) {
        const int k0 = t * BK;
        const bool kfull = k0 + BK <= g.k;
        if (fullA && kfull) load_tile<BM, THREADS, true>(g.A, g.lda, row0, k0, g.m, g.k, tid, ra);
        else                load_tile<BM, THREADS, false>(g.A, g.lda, row0, k0, g.m, g.k, tid, ra);
        if (fullB && kfull) load_tile<BN, THREADS, true>(g.B, g.ldb, col0, k0, g.n, g.k, tid, rb);
        else                load_tile<BN, THREADS, false>(g.B, g.ldb, col0, k0, g.n, g.k, tid, rb);
    };

    if (T > 0) {
        fetch(0);
        store_tile<BM, THREADS>(sA0, tid, ra);
        store_tile<BN, THREADS>(sB0, tid, rb);
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        if (t + 1 < T) fetch(t + 1);
        const double *sA = sA0 + cur * BK * LDA_S + wm * 64 + l15;
        const double *sB = sB0 + cur * BK * LDB_S + wn * 64 + l15;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = sB[(kk * 4 + l4) * LDB_S + i * 16];  // MFMA A operand <- n side
                fb[i] = sA[(kk * 4 + l4) * LDA_S + i * 16];  // MFMA B operand <- m side
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < T) {
            store_tile<BM, THREADS>(sA0 + (cur ^ 1) * BK * LDA_S, tid, ra);
            store_tile<BN, THREADS>(sB0 + (cur ^ 1) * BK * LDB_S, tid, rb);
        }
        __syncthreads();
    }

    // epilogue: acc[i][j][r] is C(m = row0 + wm*64 + j*16 + l15, n = col0 + wn*64 + i*16 + 4r + l4)
    const double alpha = g.alpha, beta = g.beta;
    const bool interior = (row0 + BM <= g.m) && (col0 + BN <= g.n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = col0 + wn * 64 + i * 16 + 4 * r + l4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = row0 + wm * 64 + j * 16 + l15;
                if (interior || (m < g.m && n < g.n)) {
                    double *c = g.C + (size_t)m + (size_t)n * g.ldc;
                    const double v = alpha * acc[i][j][r];
                    *c = (beta == 0.0) ? v : __builtin_fma(beta, *c, v);
                }
            }
        }
    }
}

}  // namespace

int gemm_nt(int m, int n, int k, double alpha, const double *A, size_t lda, const double *B,
            size_t ldb, double beta, double *C, size_t ldc, int lower, long diag_off,
            hipStream_t st)
{
    if (m < 0 || n < 0 || k < 0) { set_error("gemm_nt: negative extent"); return SGPR_E_ARG; }
    if (m == 0 || n == 0) return 0;
    if (lda < (size_t)m || ldb < (size_t)n || ldc < (size_t)m) {
        set_error("gemm_nt: leading dimension too small");
        return SGPR_E_ARG;
    }
    GemmArgs g{m, n, k, alpha, beta, A, lda, B, ldb, C, ldc, lower, diag_off};
    ProfRec rec{};
    if (g_prof.on) {
        // algorithmic flop of this launch: 2k per updated element (lower: on/below the diagonal)
        double elems = (double)m * n;
        if (lower) {
            elems = 0.0;
            for (int j = 0; j < n; ++j) {
                long first = (long)j - diag_off;  // first row with row + diag_off >= col
                if (first < 0) first = 0;
                if (first < m) elems += (double)(m - first);
            }
        }
        rec.flop = 2.0 * k * elems;
        SGPR_HIP(hipEventCreate(&rec.a));
        SGPR_HIP(hipEventCreate(&rec.b));
        SGPR_HIP(hipEventRecord(rec.a, st));
    }
    // big tile once it yields enough workgroups to fill 256 CUs, small tile below that
    const long big = (long)((m + 255) / 256) * ((n + 127) / 128);
    if (big >= 256 || (m >= 256 && n == 128)) {
        const dim3 grid((m + 255) / 256, (n + 127) / 128);
        if (grid.y > 65535) { set_error("gemm_nt: n too large for one launch"); return SGPR_E_ARG; }
        hipLaunchKernelGGL((gemm_nt_kernel<256, 128>), grid, dim3(512), 0, st, g);
        rec.big = 1;
    } else {
        const dim3 grid((m + 127) / 128, (n + 127) / 128);
        hipLaunchKernelGGL((gemm_nt_kernel<128, 128>), grid, dim3(256), 0, st, g);
    }
    SGPR_CHECK_LAUNCH();
    if (g_prof.on) {
        SGPR_HIP(hipEventRecord(rec.b, st));
        g_prof.recs.push_back(rec);
    }
    return 0;
}

void gemm_profile_begin()
{
    for (auto &r : g_prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.recs.clear();
    g_prof.on = true;
}

// out[0..2]: big-tile launches / flop / ms; out[3..5]: small-tile; out[6..7]: largest launch flop / ms
int gemm_profile_end(double *out)
{
    g_prof.on = false;
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    for (auto &r : g_prof.recs) {
        SGPR_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        SGPR_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        const int o = r.big ? 0 : 3;
        out[o] += 1.0; out[o + 1] += r.flop; out[o + 2] += ms;
        if (r.flop > out[6]) { out[6] = r.flop; out[7] = ms; }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.recs.clear();
    return 0;
}

}  // namespace sgpr

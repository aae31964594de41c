// probe.hip -- two calibration micro-benchmarks for the roofline denominators (SURVEY.md 8(d):
// "verify both peaks on the box"): a register-only fp64 MFMA issue loop and a streaming
// 16-B/lane HBM write.  Measurement aids only; nothing in the fit path calls them.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "cholq.h"
#include "common.h"
#include "../../include/sympgpr_probe.h"
#include "generated/pair_generated.h"

namespace sgpr {
namespace {
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// stamps[4*block + {0,1,2,3}] = shader-clock start/end, 100 MHz real-time start/end (wave 0)
template <int NACC>
__global__ __launch_bounds__(256) void mfma_clock_kernel(int iters, double *out, unsigned long long *stamps)
{
    double4_t acc[NACC];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)  // asm: keeps hipcc from shuttling the accumulators VGPR<->AGPR
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0) {
        stamps[4 * blockIdx.x + 0] = t0; stamps[4 * blockIdx.x + 1] = t1;
        stamps[4 * blockIdx.x + 2] = r0; stamps[4 * blockIdx.x + 3] = r1;
    }
}

__global__ __launch_bounds__(256) void mfma_probe_kernel(int iters, double *out)
{
    double4_t acc[8];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;  // keep the chain alive
}

__global__ __launch_bounds__(256) void write_probe_kernel(double2_t *dst, size_t n2)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const double2_t v{1.0, 2.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) dst[i] = v;
}
}  // namespace
}  // namespace sgpr

using namespace sgpr;

extern "C" int sgpr_probe_mfma_f64(int waves_per_simd, int iters, double *tflops)
{
    double *d = nullptr;
    SGPR_HIP(hipMalloc((void **)&d, 64));
    hipEvent_t a, b;
    SGPR_HIP(hipEventCreate(&a));
    SGPR_HIP(hipEventCreate(&b));
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks: one wave per SIMD each
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, nullptr, iters / 10 + 1, d);
    SGPR_HIP(hipEventRecord(a, nullptr));
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, nullptr, iters, d);
    SGPR_HIP(hipEventRecord(b, nullptr));
    SGPR_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    SGPR_HIP(hipEventElapsedTime(&ms, a, b));
    *tflops = (double)blocks * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipFree(d);
    return 0;
}

// out3: [0] TFLOP/s (event-timed), [1] shader cycles per MFMA per SIMD (median block),
//       [2] shader clock in GHz held during the loop (median block)
extern "C" int sgpr_probe_mfma_clock(int nacc, int waves_per_simd, int iters, double *out3)
{
    const int blocks = 256 * waves_per_simd;
    double *d = nullptr;
    unsigned long long *st = nullptr;
    SGPR_HIP(hipMalloc((void **)&d, 64));
    SGPR_HIP(hipMalloc((void **)&st, sizeof(unsigned long long) * 4 * blocks));
    hipEvent_t a, b;
    SGPR_HIP(hipEventCreate(&a));
    SGPR_HIP(hipEventCreate(&b));
    auto launch = [&](int it) {
        if (nacc == 4) hipLaunchKernelGGL((mfma_clock_kernel<4>), dim3(blocks), dim3(256), 0, nullptr, it, d, st);
        else if (nacc == 16) hipLaunchKernelGGL((mfma_clock_kernel<16>), dim3(blocks), dim3(256), 0, nullptr, it, d, st);
        else hipLaunchKernelGGL((mfma_clock_kernel<8>), dim3(blocks), dim3(256), 0, nullptr, it, d, st);
    };
    if (nacc != 4 && nacc != 16) nacc = 8;
    launch(iters);  // warm / ramp the clocks
    SGPR_HIP(hipEventRecord(a, nullptr));
    launch(iters);
    SGPR_HIP(hipEventRecord(b, nullptr));
    SGPR_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    SGPR_HIP(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(4 * (size_t)blocks);
    SGPR_HIP(hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> cyc(blocks), ghz(blocks);
    for (int i = 0; i < blocks; ++i) {
        const double dc = (double)(h[4 * i + 1] - h[4 * i]), dr = (double)(h[4 * i + 3] - h[4 * i + 2]);
        cyc[i] = dc / ((double)iters * nacc) / waves_per_simd;  // per SIMD: waves_per_simd waves share the pipe
        ghz[i] = dr > 0 ? dc / (dr * 10.0) : 0.0;              // 100 MHz real-time ticks -> 10 ns each
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    out3[0] = (double)blocks * 4 * iters * nacc * 2048.0 / (ms * 1e-3) / 1e12;
    out3[1] = cyc[blocks / 2];
    out3[2] = ghz[blocks / 2];
    (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipFree(d); (void)hipFree(st);
    return 0;
}

extern "C" int sgpr_probe_hbm_write(size_t bytes, int reps, double *gbs)
{
    double *d = nullptr;
    SGPR_HIP(hipMalloc((void **)&d, bytes));
    hipEvent_t a, b;
    SGPR_HIP(hipEventCreate(&a));
    SGPR_HIP(hipEventCreate(&b));
    hipLaunchKernelGGL(write_probe_kernel, dim3(2048), dim3(256), 0, nullptr, (double2_t *)d, bytes / 16);
    SGPR_HIP(hipEventRecord(a, nullptr));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(write_probe_kernel, dim3(2048), dim3(256), 0, nullptr, (double2_t *)d, bytes / 16);
    SGPR_HIP(hipEventRecord(b, nullptr));
    SGPR_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    SGPR_HIP(hipEventElapsedTime(&ms, a, b));
    *gbs = (double)bytes * reps / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipFree(d);
    return 0;
}

// Diagnostic: run one C -= A B^T (m x n x k, synthetic operands) with per-workgroup stamps and
// report [0] TFLOP/s (event), [1] median shader cycles a workgroup spent in its k-loop,
// [2] median shader clock (GHz) during it, [3] k-steps (of 16) per workgroup.
static thread_local int t_probe_dbg = 0;   // switches of the NEXT sgpr_probe_gemm calls on this thread only
extern "C" int sgpr_probe_gemm_debug(int bits) { t_probe_dbg = bits; return 0; }

extern "C" int sgpr_probe_gemm(int m, int n, int k, int lower, double *out4)
{
    double *A = nullptr, *B = nullptr, *Cm = nullptr;
    unsigned long long *st = nullptr;
    // leading-dimension padding (doubles) of all three operands: SGPR_PROBE_PAD, default 0
    const char *pe = getenv("SGPR_PROBE_PAD");
    const size_t pad = pe ? (size_t)atoi(pe) : 0;
    const size_t lda = (size_t)m + pad, ldb = (size_t)n + pad, ldc = (size_t)m + pad;
    SGPR_HIP(hipMalloc((void **)&A, sizeof(double) * lda * k));
    SGPR_HIP(hipMalloc((void **)&B, sizeof(double) * ldb * k));
    SGPR_HIP(hipMalloc((void **)&Cm, sizeof(double) * ldc * n));
    const size_t nwg = (size_t)((m + 127) / 128 + 8) * ((n + 127) / 128 + 8);
    SGPR_HIP(hipMalloc((void **)&st, sizeof(unsigned long long) * 4 * nwg));
    SGPR_HIP(hipMemset(st, 0, sizeof(unsigned long long) * 4 * nwg));
    // random-ish operands (not zeros: DVFS reads high on trivial data)
    KConst kc;
    const double hyp[3] = {0.7, 0.9, 1.0};
    make_kconst(SGPR_FAM_C, hyp, 3, &kc);
    std::vector<double> h((size_t)std::max(m, n) + k);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.37 * (double)((i * 2654435761u) % 1000) / 100.0;
    double *pts = nullptr;
    SGPR_HIP(hipMalloc((void **)&pts, sizeof(double) * h.size()));
    SGPR_HIP(hipMemcpy(pts, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    gram_reg(SGPR_FAM_C, m, k, pts, pts, pts + 3, pts + 5, kc, A, lda, 0, 0.0, nullptr);
    gram_reg(SGPR_FAM_C, n, k, pts, pts + 1, pts + 2, pts + 7, kc, B, ldb, 0, 0.0, nullptr);
    SGPR_HIP(hipMemset(Cm, 0, sizeof(double) * ldc * n));
    hipEvent_t a, b;
    SGPR_HIP(hipEventCreate(&a));
    SGPR_HIP(hipEventCreate(&b));
    const char *be = getenv("SGPR_PROBE_BETA");   // 0: no read of C in the epilogue
    const double beta = be ? atof(be) : 1.0;
    int rc = gemm_nt_diag(m, n, k, -1.0, A, lda, B, ldb, beta, Cm, ldc, lower, nullptr, t_probe_dbg, nullptr);  // warm
    if (rc) return rc;
    SGPR_HIP(hipEventRecord(a, nullptr));
    rc = gemm_nt_diag(m, n, k, -1.0, A, lda, B, ldb, beta, Cm, ldc, lower, st, t_probe_dbg, nullptr);
    SGPR_HIP(hipEventRecord(b, nullptr));
    if (rc) return rc;
    SGPR_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    SGPR_HIP(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> hs(4 * nwg);
    SGPR_HIP(hipMemcpy(hs.data(), st, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> cyc, ghz;
    for (size_t i = 0; i < nwg; ++i) {
        if (hs[4 * i + 1] == 0) continue;
        const double dc = (double)(hs[4 * i + 1] - hs[4 * i]), dr = (double)(hs[4 * i + 3] - hs[4 * i + 2]);
        cyc.push_back(dc);
        if (dr > 0) ghz.push_back(dc / (dr * 10.0));
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    double flop = 2.0 * m * n * (double)k;
    if (lower) flop *= 0.5;
    out4[0] = flop / (ms * 1e-3) / 1e12;
    out4[1] = cyc.empty() ? 0 : cyc[cyc.size() / 2];
    out4[2] = ghz.empty() ? 0 : ghz[ghz.size() / 2];
    out4[3] = (k + 15) / 16;
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(Cm); (void)hipFree(st); (void)hipFree(pts);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 0;
}

// Diagnostic: phase cycle counts of one 128x128 leaf factorisation (out8, shader cycles)

// Latency census of the building blocks of the leaf's 16 x 16 diagonal-block step (leaf.h, (A)): each variant runs 16
// dependent "column steps" on ONE wave (the other three of the workgroup idle, as in the leaf) between two s_memtime.
namespace sgpr { namespace {
template <int V>
__global__ __launch_bounds__(256) void lat_probe_kernel(double *io, unsigned long long *cyc)
{
    __shared__ double sh[64];
    __shared__ double big[16 * 130];
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    if (threadIdx.x >= 64) return;
    double mneg[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) mneg[j] = (l4 == (j & 3) && l15 > j) ? -1.0 : 0.0;
    double4_t D, X;
#pragma unroll
    for (int r = 0; r < 4; ++r) { D[r] = (4 * r + l4 == l15) ? 4.0 + 0.01 * l15 : 1.0 / (2.0 + l15 + 4 * r + l4); X[r] = (4 * r + l4 == l15) ? 1.0 : 0.0; }
    double rl = io[0], acc = io[1];
    auto bcast = [&](double v, int ln) {
        const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(v), ln);
        const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(v), ln);
        return __hiloint2double((int)hi, (int)lo);
    };
    auto nr = [&](double d) {
        double r = __builtin_amdgcn_rsq(d);
        r = r * __builtin_fma(-0.5 * d * r, r, 1.5);
        r = r * __builtin_fma(-0.5 * d * r, r, 1.5);
        return r;
    };
    unsigned long long t0, t1;
    // (the stamps are tied to the data flow: the loop's inputs pass through the first asm, its results into the second)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(rl), "+v"(acc), "+v"(D), "+v"(X) : : "memory");
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int rj = j >> 2;
        if (V == 0) {            // one dependent MFMA per step, operand from the accumulator (scaled)
            const double v = D[rj] * rl;
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, D, 0, 0, 0);
        } else if (V == 1) {     // two MFMAs per step (second independent of the first)
            const double v = D[rj] * rl, x = X[rj] * rl;
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, D, 0, 0, 0);
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(v, x, X, 0, 0, 0);
        } else if (V == 2) {     // the pivot chain alone: 2 readlane pairs, fma, rsq, two Newton steps
            const double a = bcast(acc, (j + 1) & 63), b = bcast(rl, (j + 17) & 63);
            rl = nr(__builtin_fma(-b, b, a + 5.0));
            acc += rl;
        } else if (V == 3) {     // rsq alone, dependent
            rl = __builtin_amdgcn_rsq(rl + 1.0);
        } else if (V == 4) {     // eight dependent DP fma
#pragma unroll
            for (int q = 0; q < 8; ++q) rl = __builtin_fma(rl, acc, 0.5);
        } else if (V == 5) {     // readlane pair -> VALU use, dependent
            rl = bcast(rl, (j + 1) & 63) + acc;
        } else if (V == 6) {     // LDS write -> read round trip, same wave
            sh[lane] = rl;
            rl = sh[(lane + 1) & 63] + 1.0;
        } else if (V >= 8) {     // the leaf's loop: 8 = as it is, 9 = without the identity tile, 10 = without LDS writes, 11 = 9 without LDS writes
            const int g = j & 3;
            const double v = D[rj] * rl;
            const double x = X[rj] * rl;
            double rn = 0.0;
            if (j + 1 < 16) {
                const double dnext = bcast(D[(j + 1) >> 2], (j + 1) + 16 * ((j + 1) & 3));
                const double lnext = bcast(v, (j + 1) + 16 * g);
                rn = nr(__builtin_fma(-lnext, lnext, dnext));
            }
            if (V == 8 || V == 9) {
                if (l4 == g) big[j * 130 + l15] = v;
                sh[j] = rl;
            }
            const double nv = v * mneg[j];
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(nv, v, D, 0, 0, 0);
            if (V == 8 || V == 10) X = __builtin_amdgcn_mfma_f64_16x16x4f64(nv, x, X, 0, 0, 0);
            rl = rn;
        } else if (V == 7) {     // MFMA with result read through a readlane and back into the next operand
            const double v = D[rj] * rl;
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, D, 0, 0, 0);
            rl = bcast(D[0], 0) * 1e-3;
        }
    }
    double ssum = rl + acc + D[0] + D[1] + D[2] + D[3] + X[0] + X[1] + X[2] + X[3];
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(ssum) : : "memory");
    if (lane == 0) cyc[V] = t1 - t0;
    if (ssum == 12345.678) io[2] = ssum + big[lane] + sh[lane];
}
} }

extern "C" int sgpr_probe_lat(double *out8 /* 12 values */)
{
    double *io = nullptr;
    unsigned long long *cyc = nullptr;
    SGPR_HIP(hipMalloc((void **)&io, 64));
    SGPR_HIP(hipMalloc((void **)&cyc, 128));
    const double h[3] = {0.49, 1.25, 0.0};
    SGPR_HIP(hipMemcpy(io, h, sizeof(h), hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<0>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<1>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<2>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<3>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<4>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<5>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<6>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<7>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<8>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<9>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<10>), dim3(1), dim3(256), 0, 0, io, cyc);
        hipLaunchKernelGGL((sgpr::lat_probe_kernel<11>), dim3(1), dim3(256), 0, 0, io, cyc);
        SGPR_HIP(hipDeviceSynchronize());
    }
    unsigned long long hs[12];
    SGPR_HIP(hipMemcpy(hs, cyc, 96, hipMemcpyDeviceToHost));
    for (int i = 0; i < 12; ++i) out8[i] = (double)hs[i] / 16.0;
    (void)hipFree(io); (void)hipFree(cyc);
    return 0;
}

extern "C" int sgpr_probe_leaf(double *out8)
{
    const int n = LEAF;
    std::vector<double> h((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) h[i + (size_t)j * n] = (i == j) ? 4.0 + 0.01 * i : 1.0 / (1.0 + i + j);
    double *A = nullptr, *inv = nullptr;
    int *info = nullptr;
    unsigned long long *st = nullptr;
    SGPR_HIP(hipMalloc((void **)&A, sizeof(double) * n * n));
    SGPR_HIP(hipMalloc((void **)&inv, sizeof(double) * n * n));
    SGPR_HIP(hipMalloc((void **)&info, 16));
    SGPR_HIP(hipMalloc((void **)&st, 64));
    SGPR_HIP(hipMemset(info, 0, 16));
    for (int rep = 0; rep < 2; ++rep) {
        SGPR_HIP(hipMemcpy(A, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
        int rc = leaf_probe(A, n, inv, info, st, nullptr);
        if (rc) return rc;
        SGPR_HIP(hipDeviceSynchronize());
    }
    unsigned long long hs[8];
    SGPR_HIP(hipMemcpy(hs, st, 64, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out8[i] = (double)hs[i];
    (void)hipFree(A); (void)hipFree(inv); (void)hipFree(info); (void)hipFree(st);
    return 0;
}

// Diagnostic: which XCD does workgroup b of a 1-D grid land on?  out[b] = HW_REG_XCC_ID of block b
// (512-thread blocks, like the MFMA kernel).  Used to check the `id % 8` assumption of the tile map.
namespace sgpr { namespace {
__global__ __launch_bounds__(512) void xcc_probe_kernel(int *out)
{
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    const int x = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
    // keep the block alive for a while so that all CUs fill up like a real launch
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}
} }
extern "C" int sgpr_probe_xcc(int nblocks, int *host_out)
{
    int *d = nullptr;
    SGPR_HIP(hipMalloc((void **)&d, sizeof(int) * nblocks));
    hipLaunchKernelGGL(sgpr::xcc_probe_kernel, dim3(nblocks), dim3(512), 0, nullptr, d);
    SGPR_HIP(hipMemcpy(host_out, d, sizeof(int) * nblocks, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return 0;
}

// Diagnostic: where do the workgroups of a CU-masked stream run?  out[2b] = XCC id, out[2b+1] = HW_ID
// (se/sh/cu fields) of block b, launched on a stream created with hipExtStreamCreateWithCUMask.
namespace sgpr { namespace {
__global__ __launch_bounds__(512) void cumask_probe_kernel(int *out)
{
    const int x = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    const int hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = hw; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) {}
}
} }
extern "C" int sgpr_probe_cumask(const unsigned *mask_words, int nwords, int nblocks, int *host_out)
{
    hipStream_t st = nullptr;
    SGPR_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask_words));
    int *d = nullptr;
    SGPR_HIP(hipMalloc((void **)&d, sizeof(int) * 2 * nblocks));
    hipLaunchKernelGGL(sgpr::cumask_probe_kernel, dim3(nblocks), dim3(512), 0, st, d);
    SGPR_HIP(hipStreamSynchronize(st));
    SGPR_HIP(hipMemcpy(host_out, d, sizeof(int) * 2 * nblocks, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    (void)hipStreamDestroy(st);
    return 0;
}


// ---- the generator's output, evaluated as it stands (tests/test_gpu_generated.py diffs the hand-optimised
// kernels of pair_eval.h against it)
namespace sgpr {
template <int FAM>
__global__ void generated_eval_kernel(int which, int m, const double *xa, const double *ya, const double *xb,
                                      const double *yb, double lx, double ly, double p, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double o[4];
    const int dl = which >> 2;
    if (dl == 0)      gen::pair<FAM>(xa[i], ya[i], xb[i], yb[i], lx, ly, p, o);
    else if (dl == 1) gen::pair_dlx<FAM>(xa[i], ya[i], xb[i], yb[i], lx, ly, p, o);
    else              gen::pair_dly<FAM>(xa[i], ya[i], xb[i], yb[i], lx, ly, p, o);
    out[i] = o[which & 3];
}
}  // namespace sgpr

extern "C" int sgpr_probe_generated_eval(int family, int which, int m, const double *xa, const double *ya,
                                         const double *xb, const double *yb, const double *l, int nl, double *out)
{
    using namespace sgpr;
    if (m <= 0) return 0;
    if (which < 0 || which > 11 || !l || nl < 2 || (family == SGPR_FAM_D && nl < 3)) { set_error("generated_eval: bad arguments"); return SGPR_E_ARG; }
    double *d[5] = {};
    const double *h[4] = {xa, ya, xb, yb};
    for (int k = 0; k < 5; ++k) SGPR_HIP(hipMalloc((void **)&d[k], sizeof(double) * m));
    for (int k = 0; k < 4; ++k) SGPR_HIP(hipMemcpy(d[k], h[k], sizeof(double) * m, hipMemcpyHostToDevice));
    const double p = family == SGPR_FAM_D ? l[2] : 0.0;
    const dim3 grid((m + 255) / 256);
    switch (family) {
    case SGPR_FAM_A: hipLaunchKernelGGL(generated_eval_kernel<SGPR_FAM_A>, grid, dim3(256), 0, nullptr, which, m, d[0], d[1], d[2], d[3], l[0], l[1], p, d[4]); break;
    case SGPR_FAM_B: hipLaunchKernelGGL(generated_eval_kernel<SGPR_FAM_B>, grid, dim3(256), 0, nullptr, which, m, d[0], d[1], d[2], d[3], l[0], l[1], p, d[4]); break;
    case SGPR_FAM_C: hipLaunchKernelGGL(generated_eval_kernel<SGPR_FAM_C>, grid, dim3(256), 0, nullptr, which, m, d[0], d[1], d[2], d[3], l[0], l[1], p, d[4]); break;
    case SGPR_FAM_D: hipLaunchKernelGGL(generated_eval_kernel<SGPR_FAM_D>, grid, dim3(256), 0, nullptr, which, m, d[0], d[1], d[2], d[3], l[0], l[1], p, d[4]); break;
    default: set_error("unknown kernel family"); return SGPR_E_ARG;
    }
    SGPR_CHECK_LAUNCH();
    SGPR_HIP(hipMemcpy(out, d[4], sizeof(double) * m, hipMemcpyDeviceToHost));
    for (int k = 0; k < 5; ++k) (void)hipFree(d[k]);
    return 0;
}

// ---- task-queue Cholesky diagnostics (cholq.h)
extern "C" int sgpr_probe_queue_plan(int n, int nworkers, int *starts_out, int max_starts, unsigned *tasks_out,
                                     int max_tasks, int *counts)
{
    cholq::Plan p;
    const int rc = cholq::build_plan(n, cholq::default_starts(n), nworkers, p);
    if (rc) return rc;
    counts[0] = p.nblk;
    counts[1] = (int)(p.tasks.size() / 2);
    counts[2] = (int)p.model_us;
    for (int k = 0; k <= p.nblk && k < max_starts; ++k) starts_out[k] = p.starts[k];
    for (size_t t = 0; t < p.tasks.size() && (int)(t / 2) < max_tasks; ++t) tasks_out[t] = p.tasks[t];
    return 0;
}

// the same with the hand-over point chosen: nq panels for the queue (< 0: the default of this order); counts[3] = nq used
extern "C" int sgpr_probe_queue_plan_partial(int n, int nworkers, int nq, int *starts_out, int max_starts, unsigned *tasks_out,
                                             int max_tasks, int *counts)
{
    cholq::Plan p;
    const std::vector<int> st = cholq::default_starts(n);
    const int rc = cholq::build_plan(n, st, nworkers, p, nq < 0 ? cholq::default_nq(n, st) : nq);
    if (rc) return rc;
    counts[0] = p.nblk;
    counts[1] = (int)(p.tasks.size() / 2);
    counts[2] = (int)p.model_us;
    counts[3] = p.nq;
    for (int k = 0; k <= p.nblk && k < max_starts; ++k) starts_out[k] = p.starts[k];
    for (size_t t = 0; t < p.tasks.size() && (int)(t / 2) < max_tasks; ++t) tasks_out[t] = p.tasks[t];
    return 0;
}

static unsigned long long *g_qtrace = nullptr;
static int g_qtrace_cap = 0;
extern "C" int sgpr_probe_queue_trace_begin(int max_tasks)
{
    if (g_qtrace) { cholq::set_trace(nullptr, 0); (void)hipFree(g_qtrace); g_qtrace = nullptr; }
    SGPR_HIP(hipMalloc((void **)&g_qtrace, sizeof(unsigned long long) * cholq::trace_words((size_t)max_tasks)));
    SGPR_HIP(hipMemset(g_qtrace, 0, sizeof(unsigned long long) * cholq::trace_words((size_t)max_tasks)));
    g_qtrace_cap = max_tasks;
    cholq::set_trace(g_qtrace, (size_t)max_tasks);
    return max_tasks;
}
extern "C" int sgpr_probe_queue_trace_clear()
{
    if (!g_qtrace) return 0;
    SGPR_HIP(hipMemset(g_qtrace, 0, sizeof(unsigned long long) * cholq::trace_words((size_t)g_qtrace_cap)));
    return 0;
}
extern "C" int sgpr_probe_queue_trace_end(unsigned long long *out, int max_tasks)
{
    if (!g_qtrace) return 0;
    SGPR_HIP(hipDeviceSynchronize());
    cholq::set_trace(nullptr, 0);
    if (max_tasks < g_qtrace_cap) { set_error("queue_trace_end: buffer smaller than the capacity given to _begin"); return SGPR_E_ARG; }
    const size_t words = cholq::trace_words((size_t)g_qtrace_cap);
    SGPR_HIP(hipMemcpy(out, g_qtrace, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost));
    (void)hipFree(g_qtrace);
    g_qtrace = nullptr;
    return (int)words;
}

extern "C" int sgpr_probe_queue_postmortem(int always) { return cholq::postmortem(always != 0); }
extern "C" int sgpr_probe_queue_force_giveup(int on) { cholq::force_giveup(on); return 0; }
extern "C" int sgpr_probe_tune(const char *name, double value)
{
    if (!name) { set_error("probe_tune: null name"); return SGPR_E_ARG; }
    tune_set(name, value);
    return 0;
}
extern "C" unsigned sgpr_probe_map_calls(void) { return applymap_last_calls(); }
extern "C" int sgpr_probe_map_team(int ntest, int n0) { return applymap_team(ntest, n0); }
extern "C" int sgpr_probe_trsm_piece(int ticket, int cap, int strips, int out[5]) { if (ticket < 0 || cap < 1 || strips < 0 || !out) return SGPR_E_ARG; trsm_piece_of(ticket, cap, strips, out); return 0; }
extern "C" int sgpr_probe_trsm_counts(int strips, int cap, unsigned long long out[2])
{
    if (strips < 0 || cap < 1 || !out) return SGPR_E_ARG;
    size_t c[2];
    trsm_piece_counts(strips, cap, c);
    out[0] = c[0]; out[1] = c[1];
    return 0;
}

// ---- co-residency census: where and when do the workgroups of two concurrent kernels run?
// Kernel A (grid na, lds_a bytes of dynamic LDS, spins spin_a us) on one stream, kernel B (nb, lds_b, spin_b) on
// another, started 50 us later.  out: per workgroup (A first, then B) 4 words: XCC id, HW_ID, start, end (100 MHz
// real time).  mask_a / mask_b: CU masks of the two streams (nwords = 0: unmasked).
namespace sgpr { namespace {
__global__ void census_kernel(unsigned long long *out, int spin_us)
{
    extern __shared__ double dyn[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        dyn[0] = 1.0;
        out[4 * blockIdx.x + 0] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        out[4 * blockIdx.x + 1] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        out[4 * blockIdx.x + 2] = t0;
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) out[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
}
} }
extern "C" int sgpr_probe_census(int na, int threads_a, int lds_a, int spin_a, const unsigned *mask_a, int nwords_a,
                                 int nb, int threads_b, int lds_b, int spin_b, const unsigned *mask_b, int nwords_b,
                                 unsigned long long *host_out)
{
    hipStream_t sa = nullptr, sb = nullptr;
    if (nwords_a > 0) SGPR_HIP(hipExtStreamCreateWithCUMask(&sa, (uint32_t)nwords_a, mask_a));
    else              SGPR_HIP(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    if (nwords_b > 0) SGPR_HIP(hipExtStreamCreateWithCUMask(&sb, (uint32_t)nwords_b, mask_b));
    else {
        int lo = 0, hi = 0;
        SGPR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        SGPR_HIP(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
    }
    unsigned long long *d = nullptr;
    const size_t words = 4 * (size_t)(na + nb);
    SGPR_HIP(hipMalloc((void **)&d, sizeof(unsigned long long) * words));
    SGPR_HIP(hipMemset(d, 0, sizeof(unsigned long long) * words));
    SGPR_HIP(hipFuncSetAttribute((const void *)sgpr::census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SGPR_HIP(hipDeviceSynchronize());
    hipLaunchKernelGGL(sgpr::census_kernel, dim3(na), dim3(threads_a), lds_a, sa, d, spin_a);
    SGPR_CHECK_LAUNCH();
    hipLaunchKernelGGL(sgpr::census_kernel, dim3(nb), dim3(threads_b), lds_b, sb, d + 4 * (size_t)na, spin_b);
    SGPR_CHECK_LAUNCH();
    SGPR_HIP(hipStreamSynchronize(sa));
    SGPR_HIP(hipStreamSynchronize(sb));
    SGPR_HIP(hipMemcpy(host_out, d, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    (void)hipStreamDestroy(sa);
    (void)hipStreamDestroy(sb);
    return 0;
}

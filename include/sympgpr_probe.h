/*
 * sympgpr_probe.h -- C ABI of libsympgpr_probe.so: roofline calibration probes and kernel
 * diagnostics for the MI355X build of SympGPR's training core.  MEASUREMENT AIDS ONLY: nothing
 * here replaces an interface of the reference, a maintainer binding include/sympgpr_hip.h never
 * loads this library, and no product path calls it (tools/ and bench.py's calibration do).
 * Same conventions as sympgpr_hip.h (0 = ok, < 0 = SGPR_E_*).
 */
#ifndef SYMPGPR_PROBE_H
#define SYMPGPR_PROBE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* a register-only fp64 MFMA issue loop with
 * `waves_per_simd` waves on every SIMD, and a streaming 16-B/lane write of `bytes` bytes. */
int sgpr_probe_mfma_f64(int waves_per_simd, int iters, double *tflops);
int sgpr_probe_hbm_write(size_t bytes, int reps, double *gbs);
/* out3: TFLOP/s, shader cycles per MFMA per SIMD, shader clock (GHz) held during the loop */
int sgpr_probe_mfma_clock(int nacc, int waves_per_simd, int iters, double *out3);
/* one synthetic C -= A B^T with per-workgroup stamps: TFLOP/s, median k-loop cycles per
 * workgroup, median shader clock (GHz), k-steps per workgroup */
int sgpr_probe_gemm(int m, int n, int k, int lower, double *out4);
/* switches for the following sgpr_probe_gemm calls OF THIS THREAD (nothing in the product library
 * changes): 8 = 128x128 tile shape, 16 = register-staged body; 0 = normal */
int sgpr_probe_gemm_debug(int bits);
/* shader cycles per phase of one 128x128 leaf factorisation: load, diag block, panel rows,
 * trailing update, write-back, inverse diag, inverse rows, final store */
int sgpr_probe_leaf(double *out8);
/* HW_REG_XCC_ID of each workgroup of a 1-D grid of 512-thread blocks (checks the tile map's `id % 8`) */
int sgpr_probe_xcc(int nblocks, int *host_out);
/* the same on a stream restricted by a CU mask (hipExtStreamCreateWithCUMask): out[2b] = XCC id, out[2b+1] = HW_ID */
int sgpr_probe_cumask(const unsigned *mask_words, int nwords, int nblocks, int *host_out);

/* The OUTPUT OF THE CODE GENERATOR (tools/gen_kernels.py -> csrc/generated/pair_generated.h), evaluated as
 * it stands: out[i] = f(xa[i], ya[i], xb[i], yb[i], l...) without the sig factor; `which` as in
 * sgpr_kernel_eval_host (0..3 = k, d2k/dxdx0, d2k/dydy0, d2k/dxdy0; | 4: d/dlx; | 8: d/dly).  The product
 * library's hand-optimised kernels are diffed against this in tests/test_gpu_generated.py. */
int sgpr_probe_generated_eval(int family, int which, int m, const double *xa, const double *ya,
                              const double *xb, const double *yb, const double *l, int nl, double *out);

#ifdef __cplusplus
}
#endif
#endif

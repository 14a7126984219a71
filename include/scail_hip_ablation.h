/*
 * scail_hip_ablation.h -- entry points of the MEASUREMENT build only (scail_amd/libscail_hip_abl.so, built with
 * `SCAIL_ABLATIONS=1 python -m scail_amd.build`).  The product library (include/scail_hip.h, libscail_hip.so) exports none of
 * these and contains none of the kernel variants they select; nothing under scail_amd/ outside tools/ may depend on them.
 * Used by tools/microbench.py, tools/attn4_tune.py, tools/gemm4_tune.py, tools/gemm_pmc_probe.py and by the GPU tests that
 * check the variants' parity (skipped when the measurement build is absent).
 */
#ifndef SCAIL_HIP_ABLATION_H
#define SCAIL_HIP_ABLATION_H
#include "scail_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/*
 * Select a kernel variant (same results unless stated, different schedules).  Knobs:
 *   "attn_variant"         8-wave attention family: bit 0 s_setprio around MFMA clusters, bit 1 skip no-op O rescales, bit 3
 *                          software-pipelined kernel, bits 12.. filler placement; bit 20 stamps workgroup lifetimes
 *                          (scail_debug_cycles); bits 4-5 are TIMING ABLATIONS with wrong results (no softmax / no staging)
 *   "attn4", "attn4_thr", "attn4_xcd", "attn4_kernel[:suffix]"   4-wave generated attention kernel: on/off, lazy-rescale
 *                          threshold, XCD-aware workgroup ids on/off, variant of asmgen/attn4.py variant_cfgs()
 *   "gemm_tile"            0 auto, 128, 256, 257, 260, 261 (q8), 262, 266; 1000-1599 = timing ablations (wrong results)
 *   "gemm_group_m"         q8 tile-group height
 *   "gemm4" (0 / 4 / 8), "gemm4_kernel[:suffix]"   generated GEMM kernels: off / gemm4 (the product default) / gemm8 (two waves per
 *                          SIMD, this build only); variant of asmgen/gemm4.py, gemm8.py variant_cfgs() for the bias epilogue
 *   "conv_halo"            halo-convolution layout 0-5; 10 / 11: generated conv4 kernels off / on (the product's option "conv4")
 *   "conv4_kernel[:suffix]"   variant of asmgen/conv4.py variant_cfgs() for the bias epilogue (timing ablations abl_*: wrong results;
 *                          "prof": phase timers (s_memtime) written through the residual pointer, 8 uint32 per workgroup)
 */
int scail_tune_set(const char* knob, int value);

/* out2[0] = summed workgroup lifetimes in s_memtime ticks (shader cycles), out2[1] = workgroup count of the launches made with
 * the clock-stamped variants (gemm_tile 1300-1364, attn_variant bit 20) since the last reset. */
int scail_debug_cycles(unsigned long long* out2, int reset);

#ifdef __cplusplus
}
#endif
#endif /* SCAIL_HIP_ABLATION_H */

// Sustained rate of the two dense bf16 MFMA shapes with all 256 accumulators of a wave live (one wave per SIMD, operands in
// registers, no memory traffic): does the 16x16x32 shape (half the accumulator read/write traffic per flop of 32x32x16) sustain
// a higher clock under the power limit?   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shape_probe.hip -o build/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

__global__ __launch_bounds__(256, 1) void k32(float* out, int iters) {
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(threadIdx.x * 0.001f + i); b[i][e] = (__bf16)(e * 0.01f + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2)], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 1) void k16(float* out, int iters) {
    f32x4 acc[64];
    for (int i = 0; i < 64; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(threadIdx.x * 0.001f + i); b[i][e] = (__bf16)(e * 0.01f + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 64; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], b[(i >> 3)], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 64; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    for (int rep = 0; rep < 3; ++rep)
        for (int which = 0; which < 2; ++which) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // per wave per iteration: 64 x 32x32x16 (or 128 x 16x16x32) MFMAs = 2 097 152 flop
            const double fl = (double)blocks * 4 * iters * 2097152.0;
            printf("{\"shape\": \"%s\", \"ms\": %.2f, \"TFLOPs\": %.1f}\n", which == 0 ? "32x32x16" : "16x16x32", ms, fl / ms / 1e9);
        }
    return 0;
}

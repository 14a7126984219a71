// Shared device/host helpers for libscail_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/scail_hip.h"

typedef uint16_t u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

// ---- error plumbing ---------------------------------------------------------------------------
void scail_set_error(const std::string& msg);
int scail_check_launch(const char* what);

#define SCAIL_REQUIRE(cond, msg)                                                   \
    do {                                                                           \
        if (!(cond)) {                                                             \
            scail_set_error(std::string(__func__) + ": " + (msg) + " [" #cond "]"); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

// One-time per-DEVICE setup at a call site (hipFuncSetAttribute opt-ins of dynamic LDS are per device: a process that drives several
// GPUs must repeat them on each): `static ScailDeviceOnce once_;  if (once_.need()) { ...setup...; once_.done(); }`.  Lock-free; two
// threads racing on the same device both run the (idempotent) setup.
#include <atomic>
struct ScailDeviceOnce {
    std::atomic<uint64_t> mask{0};
    static uint64_t device_bit() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        return 1ull << (dev & 63);
    }
    bool need() const { return (mask.load(std::memory_order_acquire) & device_bit()) == 0; }
    void done() { mask.fetch_or(device_bit(), std::memory_order_release); }
};

// ---- bf16 <-> f32 -----------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// round-to-nearest-even; hipcc lowers the vector convert to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ u16 f2bf(float a) {
    __bf16 r = (__bf16)a;
    return __builtin_bit_cast(u16, r);
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x);
    f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z);
    f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]);
    v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]);
    v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

// ---- activations ------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_tanh_f(float x) {
    // nn.GELU(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    // tanh(u) = 1 - 2/(exp(2u)+1); exp via exp2
    float e = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);  // 2*log2(e)
    float t = 1.0f - 2.0f / (e + 1.0f);
    return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// ---- wave / block reductions ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over a 256-thread block; `red` is 4 floats of LDS; all threads get the result
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

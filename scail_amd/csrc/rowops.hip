// Row-wise / elementwise kernels of the SCAIL DiT hot path (all HBM-bound): fused
// LayerNorm+modulate, affine LayerNorm, full-width RMSNorm (+3D RoPE), V transpose staging,
// timestep embedding, tiny linears, AdaLN tables, patchify / unpatchify, CFG+Euler.
// One 256-thread workgroup per row for the norm kernels; every bf16 access is a 16-byte vector.
#include "common.h"
#include <atomic>

#define ROW_THREADS 256
#define ROW_MAXV 3  // D <= 256 * 8 * 3 = 6144 kept in registers

// ------------------------------------------------------------------------------------------------
// LayerNorm (+ modulate | + affine)
// ------------------------------------------------------------------------------------------------
template <int MODE>  // 0: modulate (shift/scale per batch), 1: affine (w, b)
__global__ __launch_bounds__(ROW_THREADS) void ln_kernel(
    const u16* __restrict__ x, int64_t ldx, u16* __restrict__ y, int64_t ldy,
    const float* __restrict__ p0, const float* __restrict__ p1, int64_t mod_stride,
    int64_t rows_out, int64_t src_rows_per_batch, int64_t src_row_offset, int D, float eps) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const int64_t b = r / rows_out;
    const int64_t i = r - b * rows_out;
    const u16* xr = x + (b * src_rows_per_batch + src_row_offset + i) * ldx;
    u16* yr = y + r * ldy;
    const int nvec = D >> 3;
    float v[ROW_MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < ROW_MAXV; ++j) {
        const int c = threadIdx.x + j * ROW_THREADS;
        if (c < nvec) {
            uint4 u = *reinterpret_cast<const uint4*>(xr + (int64_t)c * 8);
            unpack8(u, v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];
        }
    }
    const float mean = block_sum_256(s, red) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < ROW_MAXV; ++j) {
        const int c = threadIdx.x + j * ROW_THREADS;
        if (c < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[j][e] - mean;
                q += d * d;
            }
        }
    }
    const float var = block_sum_256(q, red) / (float)D;
    const float rstd = rsqrtf(var + eps);
    const float* a0 = (MODE == 0) ? p0 + b * mod_stride : p0;  // shift | weight
    const float* a1 = (MODE == 0) ? p1 + b * mod_stride : p1;  // scale | bias
#pragma unroll
    for (int j = 0; j < ROW_MAXV; ++j) {
        const int c = threadIdx.x + j * ROW_THREADS;
        if (c < nvec) {
            float o[8];
            const float4 s0 = *reinterpret_cast<const float4*>(a0 + c * 8);
            const float4 s1 = *reinterpret_cast<const float4*>(a0 + c * 8 + 4);
            const float4 t0 = *reinterpret_cast<const float4*>(a1 + c * 8);
            const float4 t1 = *reinterpret_cast<const float4*>(a1 + c * 8 + 4);
            const float A0[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float A1[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = (v[j][e] - mean) * rstd;
                o[e] = (MODE == 0) ? n * (1.0f + A1[e]) + A0[e] : n * A0[e] + A1[e];
            }
            *reinterpret_cast<uint4*>(yr + (int64_t)c * 8) = pack8(o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// RMSNorm over D (+ optional interleaved RoPE with per-token pair tables)
// ------------------------------------------------------------------------------------------------
// Output layout: row-major (row stride ldy, slab_w == 0), or COLUMN SLABS (slab_w > 0): the D columns are cut into D / slab_w
// slabs, slab g holds a (rows, slab_w) matrix with row stride ldy (>= slab_w) at y + g * slab_stride -- the send layout of the Ulysses
// head <-> sequence all-to-all (scail_amd/parallel.py: one slab per destination rank; ldy = 3 * slab_w puts q | k | v side by side in
// ONE message per destination), written by the norm itself instead of a pack pass.
// w == nullptr: no normalisation / RoPE / scale, a plain copy into the chosen layout (the V third of the exchange).
__global__ __launch_bounds__(ROW_THREADS) void rmsnorm_rope_kernel(
    const u16* __restrict__ x, int64_t ldx, u16* __restrict__ y, int64_t ldy,
    const float* __restrict__ w, const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
    int64_t rows_per_batch, int D, int head_dim, float eps, float out_scale, int slab_w, int64_t slab_stride) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const u16* xr = x + r * ldx;
    u16* yr = y + r * ldy;
    const int nvec = D >> 3;
    auto dst = [&](int c) -> u16* {                      // 16-byte chunk c of this row in the output layout
        if (slab_w == 0) return yr + (int64_t)c * 8;
        const int col = c * 8, gsl = col / slab_w;
        return y + (int64_t)gsl * slab_stride + r * ldy + (col - gsl * slab_w);
    };
    if (w == nullptr) {
        for (int c = threadIdx.x; c < nvec; c += ROW_THREADS)
            *reinterpret_cast<uint4*>(dst(c)) = *reinterpret_cast<const uint4*>(xr + (int64_t)c * 8);
        return;
    }
    float v[ROW_MAXV][8];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < ROW_MAXV; ++j) {
        const int c = threadIdx.x + j * ROW_THREADS;
        if (c < nvec) {
            uint4 u = *reinterpret_cast<const uint4*>(xr + (int64_t)c * 8);
            unpack8(u, v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) q += v[j][e] * v[j][e];
        }
    }
    const float var = block_sum_256(q, red) / (float)D;
    const float rstd = rsqrtf(var + eps);
    const int half = head_dim >> 1;
    const int64_t tok = r % rows_per_batch;
#pragma unroll
    for (int j = 0; j < ROW_MAXV; ++j) {
        const int c = threadIdx.x + j * ROW_THREADS;
        if (c < nvec) {
            float n[8], o[8];
            const float4 w0 = *reinterpret_cast<const float4*>(w + c * 8);
            const float4 w1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
            const float W[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) n[e] = W[e] * (v[j][e] * rstd);
            if (cos_tab != nullptr) {
                // columns c*8 .. c*8+7 lie in one head; pair index inside the head:
                const int p0 = ((c * 8) % head_dim) >> 1;
                const float4 cs = *reinterpret_cast<const float4*>(cos_tab + tok * half + p0);
                const float4 sn = *reinterpret_cast<const float4*>(sin_tab + tok * half + p0);
                const float C[4] = {cs.x, cs.y, cs.z, cs.w};
                const float S[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // out[2i] = x[2i] cos - x[2i+1] sin ; out[2i+1] = x[2i+1] cos + x[2i] sin
                    o[2 * e] = n[2 * e] * C[e] - n[2 * e + 1] * S[e];
                    o[2 * e + 1] = n[2 * e + 1] * C[e] + n[2 * e] * S[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = n[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= out_scale;       // 1.0f (exact) unless the caller folds a scale into the row
            *reinterpret_cast<uint4*>(dst(c)) = pack8(o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ONE WAVE PER ROW forms of the two norm kernels (round 4): D = 512 * CPL columns, lane l holds the 16-byte chunks l, l + 64, ... of its
// row (each wave-instruction moves 1 KB of contiguous bytes), all CPL loads in flight before the first use, wave-level reductions only
// (no LDS, no barrier), four independent rows per 256-thread workgroup.  Same arithmetic as the block kernels above up to the order of
// the fp32 sums.  The block kernels had every thread wait at two (LayerNorm: four) barriers per row and left the third pass half empty
// at D = 5120 (640 chunks on 256 threads): 4.3 TB/s; see DESIGN.md section 4.3 for the measured A/B.
// ------------------------------------------------------------------------------------------------
template <int MODE, int CPL>
__global__ __launch_bounds__(256) void ln_wave_kernel(
    const u16* __restrict__ x, int64_t ldx, u16* __restrict__ y, int64_t ldy,
    const float* __restrict__ p0, const float* __restrict__ p1, int64_t mod_stride,
    int64_t rows_out, int64_t src_rows_per_batch, int64_t src_row_offset, int wgs_per_batch, float eps) {
    constexpr int D = 512 * CPL;
    // the row parameters of this workgroup's batch element in LDS: [D] multiplier (1 + scale | weight), [D] addend (shift | bias).  From
    // global memory they are 8 bytes per element against the 2 + 2 of x and y (40 KB per row through L2 at D = 5120).
    extern __shared__ __attribute__((aligned(16))) float prm[];
    const int64_t b = blockIdx.x / wgs_per_batch;
    const int g = blockIdx.x - (int)b * wgs_per_batch;
    {
        const float* a0 = (MODE == 0) ? p0 + b * mod_stride : p0;  // shift | weight
        const float* a1 = (MODE == 0) ? p1 + b * mod_stride : p1;  // scale | bias
        for (int i4 = threadIdx.x; i4 < D / 4; i4 += 256) {
            float4 m = *reinterpret_cast<const float4*>((MODE == 0 ? a1 : a0) + i4 * 4);
            const float4 ad = *reinterpret_cast<const float4*>((MODE == 0 ? a0 : a1) + i4 * 4);
            if (MODE == 0) { m.x += 1.0f; m.y += 1.0f; m.z += 1.0f; m.w += 1.0f; }
            *reinterpret_cast<float4*>(prm + i4 * 4) = m;
            *reinterpret_cast<float4*>(prm + D + i4 * 4) = ad;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)g * 4 + (threadIdx.x >> 6); i < rows_out; i += (int64_t)wgs_per_batch * 4) {
        const u16* xr = x + (b * src_rows_per_batch + src_row_offset + i) * ldx;
        u16* yr = y + (b * rows_out + i) * ldy;
        uint4 u[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) u[j] = *reinterpret_cast<const uint4*>(xr + (j * 64 + lane) * 8);
        float v[CPL][8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            unpack8(u[j], v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];
        }
        const float mean = wave_sum(s) * (1.0f / (float)D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[j][e] - mean;
                q += d * d;
            }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / (float)D) + eps);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = j * 64 + lane;
            const float4 m0 = *reinterpret_cast<const float4*>(prm + c * 8);
            const float4 m1 = *reinterpret_cast<const float4*>(prm + c * 8 + 4);
            const float4 t0 = *reinterpret_cast<const float4*>(prm + D + c * 8);
            const float4 t1 = *reinterpret_cast<const float4*>(prm + D + c * 8 + 4);
            const float Mu[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
            const float Ad[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[j][e] - mean) * rstd * Mu[e] + Ad[e];
            *reinterpret_cast<uint4*>(yr + c * 8) = pack8(o);
        }
    }
}

template <int CPL, bool HD128>
__global__ __launch_bounds__(256) void rmsnorm_rope_wave_kernel(
    const u16* __restrict__ x, int64_t ldx, u16* __restrict__ y, int64_t ldy,
    const float* __restrict__ w, const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
    int64_t rows, int64_t rows_per_batch, int head_dim, float eps, float out_scale, int slab_w, int64_t slab_stride) {
    constexpr int D = 512 * CPL;
    extern __shared__ __attribute__((aligned(16))) float prm[];       // the norm weight times out_scale: [D]
    for (int i4 = threadIdx.x; i4 < D / 4; i4 += 256) *reinterpret_cast<float4*>(prm + i4 * 4) = *reinterpret_cast<const float4*>(w + i4 * 4);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int half = head_dim >> 1;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const u16* xr = x + r * ldx;
        u16* yr = y + r * ldy;
        uint4 u[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) u[j] = *reinterpret_cast<const uint4*>(xr + (j * 64 + lane) * 8);
        // head_dim == 128: chunk j * 64 + lane starts at column (lane % 16) * 8 of its head for every j, so the lane's four (cos, sin)
        // pairs are the same for all its chunks: two 16-byte loads per row, requested with the row itself (the block kernel asks for them
        // per chunk, after the reduction: 20 loads per lane at D = 5120)
        const int64_t tok = r % rows_per_batch;
        float4 cs1 = make_float4(0.f, 0.f, 0.f, 0.f), sn1 = cs1;
        if (HD128 && cos_tab != nullptr) {
            cs1 = *reinterpret_cast<const float4*>(cos_tab + tok * 64 + (lane & 15) * 4);
            sn1 = *reinterpret_cast<const float4*>(sin_tab + tok * 64 + (lane & 15) * 4);
        }
        float v[CPL][8];
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            unpack8(u[j], v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) q += v[j][e] * v[j][e];
        }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / (float)D) + eps);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = j * 64 + lane;
            float n[8], o[8];
            const float4 w0 = *reinterpret_cast<const float4*>(prm + c * 8);
            const float4 w1 = *reinterpret_cast<const float4*>(prm + c * 8 + 4);
            const float W[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) n[e] = W[e] * (v[j][e] * rstd);
            if (cos_tab != nullptr) {
                float4 cs = cs1, sn = sn1;
                if (!HD128) {
                    const int p0 = ((c * 8) % head_dim) >> 1;
                    cs = *reinterpret_cast<const float4*>(cos_tab + tok * half + p0);
                    sn = *reinterpret_cast<const float4*>(sin_tab + tok * half + p0);
                }
                const float C[4] = {cs.x, cs.y, cs.z, cs.w};
                const float S[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[2 * e] = n[2 * e] * C[e] - n[2 * e + 1] * S[e];
                    o[2 * e + 1] = n[2 * e + 1] * C[e] + n[2 * e] * S[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = n[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= out_scale;
            u16* dst;
            if (slab_w == 0) {
                dst = yr + c * 8;
            } else {
                const int col = c * 8, gsl = col / slab_w;
                dst = y + (int64_t)gsl * slab_stride + r * ldy + (col - gsl * slab_w);
            }
            *reinterpret_cast<uint4*>(dst) = pack8(o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// V -> V^T staging (64 keys x head_dim tile through LDS)
// ------------------------------------------------------------------------------------------------
#define TV_KEYS 64
__global__ __launch_bounds__(256) void transpose_v_kernel(
    const u16* __restrict__ v, int64_t ldv, int64_t v_bs, u16* __restrict__ vt, int heads,
    int head_dim, int64_t Lk, int64_t Lkp) {
    extern __shared__ __attribute__((aligned(16))) u16 tile[];  // [64][head_dim + 8]
    const int ldt = head_dim + 8;
    const int64_t key0 = (int64_t)blockIdx.x * TV_KEYS;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int cpr = head_dim >> 3;  // 16-byte chunks per row
    const u16* src = v + b * v_bs + (int64_t)h * head_dim;
    for (int c = threadIdx.x; c < TV_KEYS * cpr; c += 256) {
        const int kr = c / cpr, cc = c - kr * cpr;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (key0 + kr < Lk) u = *reinterpret_cast<const uint4*>(src + (key0 + kr) * ldv + cc * 8);
        *reinterpret_cast<uint4*>(tile + kr * ldt + cc * 8) = u;
    }
    __syncthreads();
    u16* dst = vt + ((b * heads + h) * (int64_t)head_dim) * Lkp + key0;
    for (int o = threadIdx.x; o < head_dim * (TV_KEYS / 8); o += 256) {
        const int d = o >> 3, ch = o & 7;
        u16 e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = ch * 8 + j;  // position inside the 64-key tile
            const int key = (p & ~12) | (((p >> 3) & 1) << 2) | (((p >> 2) & 1) << 3);  // swap bits 2,3
            e[j] = tile[key * ldt + d];
        }
        uint4 u;
        u.x = e[0] | ((uint32_t)e[1] << 16);
        u.y = e[2] | ((uint32_t)e[3] << 16);
        u.z = e[4] | ((uint32_t)e[5] << 16);
        u.w = e[6] | ((uint32_t)e[7] << 16);
        *reinterpret_cast<uint4*>(dst + (int64_t)d * Lkp + ch * 8) = u;
    }
}

// ------------------------------------------------------------------------------------------------
// small stuff
// ------------------------------------------------------------------------------------------------
__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int dim) {
    const int b = blockIdx.x;
    const int half = dim >> 1;
    for (int j = threadIdx.x; j < half; j += blockDim.x) {
        const double f = exp(-log(10000.0) * (double)j / (double)half);
        const double a = (double)t[b] * f;
        out[(int64_t)b * dim + j] = (float)cos(a);
        out[(int64_t)b * dim + half + j] = (float)sin(a);
    }
    if ((dim & 1) && threadIdx.x == 0) out[(int64_t)b * dim + dim - 1] = 0.f;
}

__device__ __forceinline__ float act_f(float x, int a) {
    return a == 1 ? silu_f(x) : (a == 2 ? gelu_tanh_f(x) : x);
}

#define SL_MAXM 8
// one wave per output column n; lanes stride over K in 8-element chunks
__global__ __launch_bounds__(256) void small_linear_kernel(
    const float* __restrict__ x, const u16* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int M, int N, int K, int act_in, int act_out) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    float acc[SL_MAXM];
#pragma unroll
    for (int m = 0; m < SL_MAXM; ++m) acc[m] = 0.f;
    const u16* wr = w + (int64_t)n * K;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        float wf[8];
        unpack8(*reinterpret_cast<const uint4*>(wr + k), wf);
#pragma unroll
        for (int m = 0; m < SL_MAXM; ++m) {
            if (m < M) {
                const float4 x0 = *reinterpret_cast<const float4*>(x + (int64_t)m * K + k);
                const float4 x1 = *reinterpret_cast<const float4*>(x + (int64_t)m * K + k + 4);
                const float X[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[m] += act_f(X[e], act_in) * wf[e];
            }
        }
    }
#pragma unroll
    for (int m = 0; m < SL_MAXM; ++m) {
        if (m < M) {
            const float s = wave_sum(acc[m]);
            if (lane == 0) y[(int64_t)m * N + n] = act_f(s + (bias ? bias[n] : 0.f), act_out);
        }
    }
}

__global__ void adaln_table_kernel(const float* __restrict__ emb, const float* __restrict__ table,
                                   float* __restrict__ out, int64_t n_batch, int64_t width, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t j = i % width;
    const int64_t b = (i / width) % n_batch;
    const int64_t l = i / (width * n_batch);
    out[i] = emb[b * width + j] + table[l * width + j];
}

// one thread per (token, 8-column chunk) of the im2col matrix
__global__ void patchify_kernel(const float* __restrict__ x, const u16* __restrict__ ref,
                                const u16* __restrict__ pose, u16* __restrict__ tok, int64_t n_batch,
                                int64_t n_ref, int64_t n_pose, int T, int H, int W, int kpad, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cpr = kpad >> 3;
    const int ch = (int)(i % cpr);
    const int64_t row = i / cpr;
    const int h2 = H >> 1, w2 = W >> 1, h4 = H >> 2, w4 = W >> 2;
    const int64_t Lref = (int64_t)h2 * w2, Lnoise = (int64_t)T * h2 * w2, Lpose = (int64_t)T * h4 * w4;
    const int64_t L = Lref + Lnoise + Lpose;
    const int64_t b = row / L;
    int64_t l = row - b * L;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = ch * 8 + e;
        float val = 0.f;
        if (k < 80) {
            const int c = k >> 2, p = (k >> 1) & 1, q = k & 1;
            if (l < Lref) {
                const int hh = (int)(l / w2), ww = (int)(l % w2);
                if (c >= 16) val = 1.f;
                else {
                    const int64_t bb = (n_ref == 1) ? 0 : b;
                    val = bf2f(ref[((bb * 16 + c) * H + (2 * hh + p)) * W + 2 * ww + q]);
                }
            } else if (l < Lref + Lnoise) {
                const int64_t ll = l - Lref;
                const int t = (int)(ll / (h2 * w2));
                const int rem = (int)(ll % (h2 * w2));
                const int hh = rem / w2, ww = rem % w2;
                if (c >= 16) val = 0.f;
                else val = bf2f(f2bf(x[(((b * T + t) * 16 + c) * H + (2 * hh + p)) * W + 2 * ww + q]));
            } else {
                const int64_t ll = l - Lref - Lnoise;
                const int t = (int)(ll / (h4 * w4));
                const int rem = (int)(ll % (h4 * w4));
                const int hh = rem / w4, ww = rem % w4;
                if (c >= 16) val = 1.f;
                else {
                    const int64_t bb = (n_pose == 1) ? 0 : b;
                    val = bf2f(pose[(((bb * T + t) * 16 + c) * h2 + (2 * hh + p)) * w2 + 2 * ww + q]);
                }
            }
        }
        o[e] = val;
    }
    *reinterpret_cast<uint4*>(tok + row * kpad + ch * 8) = pack8(o);
}

// out[b][t][c][2h+p][2w+q] = tok[b][(t,h,w)][p*32 + q*16 + c]
__global__ void unpatchify_kernel(const u16* __restrict__ tok, float* __restrict__ out, int T, int H,
                                  int W, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ww = (int)(i % W);
    const int hh = (int)((i / W) % H);
    const int c = (int)((i / ((int64_t)W * H)) % 16);
    const int t = (int)((i / ((int64_t)W * H * 16)) % T);
    const int64_t b = i / ((int64_t)W * H * 16 * T);
    const int h2 = H >> 1, w2 = W >> 1;
    const int64_t row = (b * T + t) * (int64_t)(h2 * w2) + (int64_t)(hh >> 1) * w2 + (ww >> 1);
    out[i] = bf2f(tok[row * 64 + (hh & 1) * 32 + (ww & 1) * 16 + c]);
}

__global__ void cfg_euler_kernel(float* __restrict__ x, const float* __restrict__ v, int64_t n,
                                 float cfg, float dsigma) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float vu = v[i], vc = v[n + i];
    x[i] = x[i] + dsigma * (vu + cfg * (vc - vu));
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, u16* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = f2bf(x[i]);
}
__global__ void bf16_to_f32_kernel(const u16* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = bf2f(x[i]);
}

// ================================================================================================
// C ABI
// ================================================================================================
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// option "row_wave" (scail_set_option): 1 (default) = the one-wave-per-row norm kernels where D is 1536 / 2048 / 4096 / 5120 / 6144,
// 0 = the block-per-row kernels for every D (same-process A/B; the results agree up to the order of the fp32 sums)
static std::atomic<int> g_row_wave{1};   // option state: atomic (set by one thread, read by every launching thread)
int scail_row_wave_enable(int on) { g_row_wave = on != 0; return 0; }
static int row_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

// returns -1 when the wave kernels do not cover this D (the caller launches the block kernel), else the launch status
template <int MODE>
static int ln_wave_launch(const scail_bf16* x, int64_t ldx, scail_bf16* y, int64_t ldy, const float* p0, const float* p1, int64_t mod_stride,
                          int64_t rows_out, int64_t src_rows_per_batch, int64_t src_row_offset, int64_t rows, int64_t D, float eps, void* stream) {
    if (!g_row_wave || D % 512 != 0 || rows_out <= 0 || rows % rows_out != 0) return -1;
#define LN_WAVE(CPL_)                                                                                                                  \
    case CPL_: {                                                                                                                       \
        constexpr int lds_ = 2 * 512 * CPL_ * 4;                                                                                       \
        static ScailDeviceOnce attr_;                                                                                                     \
        if (attr_.need()) {                                                                                                                  \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_wave_kernel<MODE, CPL_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) { \
                scail_set_error("ln_wave: hipFuncSetAttribute failed");                                                                \
                return 2;                                                                                                              \
            }                                                                                                                          \
            attr_.done();                                                                                                              \
        }                                                                                                                              \
        hipLaunchKernelGGL((ln_wave_kernel<MODE, CPL_>), dim3((unsigned)(n_batch * wpb)), dim3(256), lds_, (hipStream_t)stream, x, ldx, y, ldy, \
                           p0, p1, mod_stride, rows_out, src_rows_per_batch, src_row_offset, (int)wpb, eps);                           \
        return scail_check_launch("ln_wave");                                                                                          \
    }
    // persistent workgroups: about three per compute unit (157 VGPRs, 40 KB of LDS), dealt evenly to the batch elements
    const int64_t n_batch = rows / rows_out;
    const int64_t wpb = std::max<int64_t>(1, std::min<int64_t>((rows_out + 3) / 4, (3 * row_cu_count() + n_batch - 1) / n_batch));
    switch (D / 512) {
        LN_WAVE(3) LN_WAVE(4) LN_WAVE(8) LN_WAVE(10) LN_WAVE(12)
        default: return -1;
    }
#undef LN_WAVE
}

extern "C" int scail_ln_modulate(const scail_bf16* x, int64_t ldx, scail_bf16* y, int64_t ldy,
                                 const float* shift, const float* scale, int64_t mod_stride,
                                 int64_t n_batch, int64_t rows_out, int64_t src_rows_per_batch,
                                 int64_t src_row_offset, int64_t D, float eps, void* stream) {
    SCAIL_REQUIRE(D % 8 == 0 && D <= ROW_THREADS * 8 * ROW_MAXV, "D must be a multiple of 8 and <= 6144");
    SCAIL_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && mod_stride % 4 == 0, "strides must keep 16-byte alignment");
    SCAIL_REQUIRE(aligned16(x) && aligned16(y) && aligned16(shift) && aligned16(scale), "pointers must be 16-byte aligned");
    const int64_t rows = n_batch * rows_out;
    if (rows == 0) return 0;
    if (int rc = ln_wave_launch<0>(x, ldx, y, ldy, shift, scale, mod_stride, rows_out, src_rows_per_batch, src_row_offset, rows, D, eps, stream); rc >= 0)
        return rc;
    hipLaunchKernelGGL(ln_kernel<0>, dim3((unsigned)rows), dim3(ROW_THREADS), 0, (hipStream_t)stream, x, ldx,
                       y, ldy, shift, scale, mod_stride, rows_out, src_rows_per_batch, src_row_offset,
                       (int)D, eps);
    return scail_check_launch("ln_modulate");
}

extern "C" int scail_layernorm_affine(const scail_bf16* x, int64_t ldx, scail_bf16* y, int64_t ldy,
                                      const float* w, const float* b, int64_t rows, int64_t D, float eps,
                                      void* stream) {
    SCAIL_REQUIRE(D % 8 == 0 && D <= ROW_THREADS * 8 * ROW_MAXV, "D must be a multiple of 8 and <= 6144");
    SCAIL_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "strides must keep 16-byte alignment");
    SCAIL_REQUIRE(aligned16(x) && aligned16(y) && aligned16(w) && aligned16(b), "pointers must be 16-byte aligned");
    if (rows == 0) return 0;
    if (int rc = ln_wave_launch<1>(x, ldx, y, ldy, w, b, 0, rows, rows, 0, rows, D, eps, stream); rc >= 0) return rc;
    hipLaunchKernelGGL(ln_kernel<1>, dim3((unsigned)rows), dim3(ROW_THREADS), 0, (hipStream_t)stream, x, ldx,
                       y, ldy, w, b, (int64_t)0, rows, rows, (int64_t)0, (int)D, eps);
    return scail_check_launch("layernorm_affine");
}

extern "C" int scail_rmsnorm_rope(const scail_bf16* x, int64_t ldx, scail_bf16* y, int64_t ldy,
                                  const float* w, const float* cos_tab, const float* sin_tab,
                                  int64_t rows, int64_t rows_per_batch, int64_t D, int64_t head_dim,
                                  float eps, void* stream) {
    return scail_rmsnorm_rope_scaled(x, ldx, y, ldy, w, cos_tab, sin_tab, rows, rows_per_batch, D, head_dim, eps, 1.0f, stream);
}

static int rmsnorm_rope_launch(const scail_bf16* x, int64_t ldx, scail_bf16* y, int64_t ldy, const float* w, const float* cos_tab,
                               const float* sin_tab, int64_t rows, int64_t rows_per_batch, int64_t D, int64_t head_dim, float eps,
                               float out_scale, int64_t slab_w, int64_t slab_stride, void* stream) {
    SCAIL_REQUIRE(D % 8 == 0 && D <= ROW_THREADS * 8 * ROW_MAXV, "D must be a multiple of 8 and <= 6144");
    SCAIL_REQUIRE(head_dim % 8 == 0 && D % head_dim == 0, "head_dim must be a multiple of 8 dividing D");
    SCAIL_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "strides must keep 16-byte alignment");
    SCAIL_REQUIRE((cos_tab == nullptr) == (sin_tab == nullptr), "cos and sin tables go together");
    SCAIL_REQUIRE(aligned16(x) && aligned16(y) && aligned16(w) && aligned16(cos_tab) && aligned16(sin_tab),
                  "pointers must be 16-byte aligned");
    SCAIL_REQUIRE(rows_per_batch > 0, "rows_per_batch must be positive");
    SCAIL_REQUIRE(slab_w == 0 || (slab_w % 8 == 0 && D % slab_w == 0 && ldy >= slab_w && slab_stride % 8 == 0 && slab_stride >= (rows - 1) * ldy + slab_w),
                  "slab width must be a multiple of 8 dividing D, slab row stride >= slab width, slab stride a multiple of 8 that holds rows of that stride");
    if (rows == 0) return 0;
    // with RoPE the wave form wins (0.455 -> 0.41 ms at config 2: the (cos, sin) pairs once per row instead of per chunk); without, the block
    // kernel is already at the pass's plateau (0.367 against 0.383 ms; profiles/r04_row_pass_probe.log)
    if (g_row_wave && w != nullptr && cos_tab != nullptr && D % 512 == 0) {
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((rows + 3) / 4, 3 * (int64_t)row_cu_count()));
#define RR_WAVE(CPL_)                                                                                                                \
    case CPL_: {                                                                                                                     \
        constexpr int lds_ = 512 * CPL_ * 4;                                                                                         \
        if (head_dim == 128)                                                                                                         \
            hipLaunchKernelGGL((rmsnorm_rope_wave_kernel<CPL_, true>), dim3(grid), dim3(256), lds_, (hipStream_t)stream, x, ldx, y,  \
                               ldy, w, cos_tab, sin_tab, rows, rows_per_batch, (int)head_dim, eps, out_scale, (int)slab_w, slab_stride); \
        else                                                                                                                         \
            hipLaunchKernelGGL((rmsnorm_rope_wave_kernel<CPL_, false>), dim3(grid), dim3(256), lds_, (hipStream_t)stream, x, ldx, y, \
                               ldy, w, cos_tab, sin_tab, rows, rows_per_batch, (int)head_dim, eps, out_scale, (int)slab_w, slab_stride); \
        return scail_check_launch("rmsnorm_rope");                                                                                   \
    }
        switch (D / 512) {
            RR_WAVE(3) RR_WAVE(4) RR_WAVE(8) RR_WAVE(10) RR_WAVE(12)
            default: break;
        }
#undef RR_WAVE
    }
    hipLaunchKernelGGL(rmsnorm_rope_kernel, dim3((unsigned)rows), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       x, ldx, y, ldy, w, cos_tab, sin_tab, rows_per_batch, (int)D, (int)head_dim, eps, out_scale, (int)slab_w, slab_stride);
    return scail_check_launch("rmsnorm_rope");
}

extern "C" int scail_rmsnorm_rope_scaled(const scail_bf16* x, int64_t ldx, scail_bf16* y, int64_t ldy,
                                         const float* w, const float* cos_tab, const float* sin_tab,
                                         int64_t rows, int64_t rows_per_batch, int64_t D, int64_t head_dim,
                                         float eps, float out_scale, void* stream) {
    SCAIL_REQUIRE(w != nullptr, "null weight");
    return rmsnorm_rope_launch(x, ldx, y, ldy, w, cos_tab, sin_tab, rows, rows_per_batch, D, head_dim, eps, out_scale, 0, 0, stream);
}

extern "C" int scail_rmsnorm_rope_slabs(const scail_bf16* x, int64_t ldx, scail_bf16* y, int64_t slab_w, int64_t slab_ld, int64_t slab_stride,
                                        const float* w, const float* cos_tab, const float* sin_tab,
                                        int64_t rows, int64_t rows_per_batch, int64_t D, int64_t head_dim,
                                        float eps, float out_scale, void* stream) {
    SCAIL_REQUIRE(slab_w > 0, "slab width must be positive");
    SCAIL_REQUIRE(w != nullptr || (cos_tab == nullptr && sin_tab == nullptr), "a plain slab copy (w == NULL) takes no RoPE tables");
    return rmsnorm_rope_launch(x, ldx, y, slab_ld, w, cos_tab, sin_tab, rows, rows_per_batch, D, head_dim, eps, out_scale, slab_w, slab_stride, stream);
}

// slabs -> rows: the way back of the Ulysses exchange (the inverse layout of scail_rmsnorm_rope_slabs): slab g is a dense [rows, slab_w]
// matrix at x + g * slab_stride; row r of y gets slab g's row r at columns [g * slab_w, (g + 1) * slab_w).  One 16-byte chunk per thread,
// chunk index fastest along the OUTPUT row (whole-line stores; the loads of a slab row are contiguous too).
__global__ void slabs_to_rows_kernel(const u16* __restrict__ x, int64_t slab_stride, int slab_chunks, u16* __restrict__ y, int64_t ldy,
                                     int row_chunks, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t r = i / row_chunks;
    const int c = (int)(i - r * row_chunks);
    const int g = c / slab_chunks, j = c - g * slab_chunks;
    const uint4 v = *reinterpret_cast<const uint4*>(x + g * slab_stride + (r * slab_chunks + j) * 8);
    *reinterpret_cast<uint4*>(y + r * ldy + (int64_t)c * 8) = v;
}

extern "C" int scail_slabs_to_rows(const scail_bf16* x, int64_t slab_w, int64_t slab_stride, scail_bf16* y, int64_t ldy,
                                   int64_t rows, int64_t D, void* stream) {
    SCAIL_REQUIRE(slab_w > 0 && slab_w % 8 == 0 && D % slab_w == 0, "slab width must be a multiple of 8 that divides D");
    SCAIL_REQUIRE(ldy % 8 == 0 && slab_stride % 8 == 0 && slab_stride >= rows * slab_w && aligned16(x) && aligned16(y), "16-byte alignment / slab stride");
    const int64_t total = rows * (D / 8);
    if (total == 0) return 0;
    SCAIL_REQUIRE(total / 256 + 1 < (1ll << 31), "too many rows");
    hipLaunchKernelGGL(slabs_to_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, slab_stride,
                       (int)(slab_w / 8), y, ldy, (int)(D / 8), total);
    return scail_check_launch("slabs_to_rows");
}

extern "C" int scail_transpose_v(const scail_bf16* v, int64_t ldv, int64_t v_batch_stride,
                                 scail_bf16* vt, int64_t n_batch, int64_t heads, int64_t head_dim,
                                 int64_t Lk, void* stream) {
    SCAIL_REQUIRE(head_dim % 8 == 0 && head_dim <= 512, "head_dim must be a multiple of 8, <= 512");
    SCAIL_REQUIRE(ldv % 8 == 0 && v_batch_stride % 8 == 0 && aligned16(v) && aligned16(vt), "16-byte alignment");
    if (Lk == 0 || n_batch == 0) return 0;
    const int64_t Lkp = (Lk + 63) / 64 * 64;
    const size_t lds = (size_t)TV_KEYS * (head_dim + 8) * sizeof(u16);
    hipLaunchKernelGGL(transpose_v_kernel, dim3((unsigned)(Lkp / 64), (unsigned)heads, (unsigned)n_batch),
                       dim3(256), lds, (hipStream_t)stream, v, ldv, v_batch_stride, vt, (int)heads,
                       (int)head_dim, Lk, Lkp);
    return scail_check_launch("transpose_v");
}

extern "C" int scail_timestep_embedding(const float* t, float* out, int64_t n, int64_t dim, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((unsigned)n), dim3(128), 0, (hipStream_t)stream, t, out,
                       (int)dim);
    return scail_check_launch("timestep_embedding");
}

extern "C" int scail_small_linear(const float* x, const scail_bf16* w, const float* b, float* y, int64_t M,
                                  int64_t N, int64_t K, int act_in, int act_out, void* stream) {
    SCAIL_REQUIRE(M >= 0 && M <= SL_MAXM, "M must be <= 8");
    SCAIL_REQUIRE(K % 8 == 0, "K must be a multiple of 8");
    SCAIL_REQUIRE(aligned16(x) && aligned16(w), "16-byte alignment");
    if (M == 0 || N == 0) return 0;
    hipLaunchKernelGGL(small_linear_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                       w, b, y, (int)M, (int)N, (int)K, act_in, act_out);
    return scail_check_launch("small_linear");
}

extern "C" int scail_adaln_table(const float* emb, const float* table, float* out, int64_t n_layers,
                                 int64_t n_batch, int64_t width, void* stream) {
    const int64_t total = n_layers * n_batch * width;
    if (total == 0) return 0;
    hipLaunchKernelGGL(adaln_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, emb, table, out, n_batch, width, total);
    return scail_check_launch("adaln_table");
}

extern "C" int scail_patchify(const float* x, const scail_bf16* ref, const scail_bf16* pose,
                              scail_bf16* tok, int64_t n_batch, int64_t n_ref, int64_t n_pose, int64_t T,
                              int64_t H, int64_t W, int64_t kpad, void* stream) {
    SCAIL_REQUIRE(H % 4 == 0 && W % 4 == 0, "latent H and W must be multiples of 4 (pose is half size, patch 2)");
    SCAIL_REQUIRE(kpad >= 80 && kpad % 8 == 0, "kpad must be >= 80 and a multiple of 8");
    SCAIL_REQUIRE((n_ref == 1 || n_ref == n_batch) && (n_pose == 1 || n_pose == n_batch), "cond batch must be 1 or n_batch");
    const int64_t L = (1 + T) * (H / 2) * (W / 2) + T * (H / 4) * (W / 4);
    const int64_t total = n_batch * L * (kpad / 8);
    if (total == 0) return 0;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, ref, pose, tok, n_batch, n_ref, n_pose, (int)T, (int)H, (int)W,
                       (int)kpad, total);
    return scail_check_launch("patchify");
}

extern "C" int scail_unpatchify(const scail_bf16* tok, float* out, int64_t n_batch, int64_t T, int64_t H,
                                int64_t W, void* stream) {
    const int64_t total = n_batch * T * 16 * H * W;
    if (total == 0) return 0;
    hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, tok, out, (int)T, (int)H, (int)W, total);
    return scail_check_launch("unpatchify");
}

extern "C" int scail_cfg_euler(float* x, const float* v, int64_t n, float cfg_scale, float dsigma,
                               void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, v, n, cfg_scale, dsigma);
    return scail_check_launch("cfg_euler");
}

extern "C" int scail_f32_to_bf16(const float* x, scail_bf16* y, int64_t n, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, y, n);
    return scail_check_launch("f32_to_bf16");
}
extern "C" int scail_bf16_to_f32(const scail_bf16* x, float* y, int64_t n, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, y, n);
    return scail_check_launch("bf16_to_f32");
}

// ================================================================================================
// Small-sequence attention for the conditioning encoders (UMT5: 64 heads x 64, relative-position bias,
// key padding mask, no scaling -- umt5.py:72-123; CLIP ViT-H: 16 heads x 80, scale 1/sqrt(80) --
// clip.py:71-108).  Sequences are <= 512 tokens and the FLOPs are negligible (4 GFLOP per T5 layer), so
// this is a plain VALU kernel: K and V of one (batch, head) live in LDS, one wave per query row.
// ================================================================================================
#define SA_ROWS 32   // query rows per workgroup (8 per wave)
__global__ __launch_bounds__(256) void attn_small_kernel(
    const u16* __restrict__ q, const u16* __restrict__ k, const u16* __restrict__ v, u16* __restrict__ o,
    int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs, int64_t o_bs, int64_t o_rs,
    int Lq, int Lk, int hd, float scale, const int* __restrict__ bucket, const float* __restrict__ bias_tab, int n_heads_tab,
    const int* __restrict__ kmask, int64_t kmask_bs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    const int ldk = hd + 8;                                  // padded rows: conflict-free 16-byte reads
    u16* Ks = reinterpret_cast<u16*>(sm_raw);                // [Lk][ldk]
    u16* Vs = Ks + (size_t)Lk * ldk;                         // [Lk][hd]
    float* Ps = reinterpret_cast<float*>(Vs + (size_t)Lk * hd);   // [4 waves][Lk]
    float* Qs = Ps + 4 * Lk;                                 // [4 waves][hd]
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpr = hd >> 3;
    for (int c = tid; c < Lk * cpr; c += 256) {
        const int j = c / cpr, cc = c - j * cpr;
        *reinterpret_cast<uint4*>(Ks + j * ldk + cc * 8) = *reinterpret_cast<const uint4*>(k + b * k_bs + (int64_t)j * k_rs + h * hd + cc * 8);
        *reinterpret_cast<uint4*>(Vs + j * hd + cc * 8) = *reinterpret_cast<const uint4*>(v + b * v_bs + (int64_t)j * v_rs + h * hd + cc * 8);
    }
    __syncthreads();
    float* P = Ps + wave * Lk;
    float* Q = Qs + wave * hd;
    const int i0 = blockIdx.x * SA_ROWS;
    for (int ii = wave; ii < SA_ROWS; ii += 4) {
        const int i = i0 + ii;
        if (i >= Lq) break;                                   // wave-uniform
        for (int d = lane; d < hd; d += 64) Q[d] = bf2f(q[b * q_bs + (int64_t)i * q_rs + h * hd + d]) * scale;
        __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): Q visible to the whole wave
        float mx = -INFINITY;
        for (int j = lane; j < Lk; j += 64) {
            float s = 0.f;
            for (int c = 0; c < cpr; ++c) {
                float kf[8];
                unpack8(*reinterpret_cast<const uint4*>(Ks + j * ldk + c * 8), kf);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += Q[c * 8 + e] * kf[e];
            }
            if (bias_tab != nullptr) s += bias_tab[bucket[(int64_t)i * Lk + j] * n_heads_tab + h];
            if (kmask != nullptr && kmask[b * kmask_bs + j] == 0) s = -INFINITY;
            P[j] = s;
            mx = fmaxf(mx, s);
        }
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) mx = fmaxf(mx, __shfl_xor(mx, of, 64));
        float sum = 0.f;
        for (int j = lane; j < Lk; j += 64) {
            const float pv = __expf(P[j] - mx);
            P[j] = pv;
            sum += pv;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const float inv = 1.0f / sum;
        for (int d = lane; d < hd; d += 64) {
            float acc = 0.f;
            for (int j = 0; j < Lk; ++j) acc += P[j] * bf2f(Vs[j * hd + d]);
            o[b * o_bs + (int64_t)i * o_rs + h * hd + d] = f2bf(acc * inv);
        }
    }
}

__global__ void mul_bf16_kernel(const u16* __restrict__ a, const u16* __restrict__ b, u16* __restrict__ y, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float fa[8], fb[8], fo[8];
    unpack8(*reinterpret_cast<const uint4*>(a + i * 8), fa);
    unpack8(*reinterpret_cast<const uint4*>(b + i * 8), fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) fo[e] = fa[e] * fb[e];
    *reinterpret_cast<uint4*>(y + i * 8) = pack8(fo);
}

// y[r, :] = x[r, :] * rowscale[r] + addrow[r % add_rows, :]   (either may be NULL)
__global__ void row_affine_kernel(const u16* __restrict__ x, u16* __restrict__ y, const float* __restrict__ rowscale,
                                  const u16* __restrict__ addrow, int64_t add_rows, int64_t rows, int D8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D8) return;
    const int64_t r = i / D8;
    const int c = (int)(i - r * D8);
    float f[8], a[8];
    unpack8(*reinterpret_cast<const uint4*>(x + i * 8), f);
    const float s = rowscale ? rowscale[r] : 1.f;
    if (addrow) unpack8(*reinterpret_cast<const uint4*>(addrow + ((r % add_rows) * D8 + c) * 8), a);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = f[e] * s + (addrow ? a[e] : 0.f);
    *reinterpret_cast<uint4*>(y + i * 8) = pack8(f);
}

extern "C" int scail_attn_small(const scail_bf16* q, const scail_bf16* k, const scail_bf16* v, scail_bf16* o,
                                const int64_t* strides, int64_t n_batch, int64_t heads, int64_t Lq, int64_t Lk, int64_t head_dim,
                                float scale, const int32_t* bucket, const float* bias_tab, const int32_t* key_mask,
                                int64_t key_mask_bs, void* stream) {
    SCAIL_REQUIRE(head_dim % 8 == 0 && head_dim <= 128, "head_dim must be a multiple of 8, <= 128");
    SCAIL_REQUIRE((bucket == nullptr) == (bias_tab == nullptr), "bucket and bias table go together");
    for (int i = 0; i < 8; ++i) SCAIL_REQUIRE(strides[i] % 8 == 0, "strides must keep 16-byte alignment");
    const size_t lds = (size_t)Lk * (head_dim + 8) * 2 + (size_t)Lk * head_dim * 2 + 4 * Lk * 4 + 4 * head_dim * 4;
    SCAIL_REQUIRE(lds <= 160 * 1024, "K and V of one head must fit in LDS (Lk * head_dim too large for this kernel)");
    if (Lq == 0 || n_batch == 0) return 0;
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { scail_set_error("attn_small: hipFuncSetAttribute failed"); return 2; }
        lds_set = lds;
    }
    dim3 grid((unsigned)((Lq + SA_ROWS - 1) / SA_ROWS), (unsigned)heads, (unsigned)n_batch);
    hipLaunchKernelGGL(attn_small_kernel, grid, dim3(256), lds, (hipStream_t)stream, q, k, v, o, strides[0], strides[1], strides[2],
                       strides[3], strides[4], strides[5], strides[6], strides[7], (int)Lq, (int)Lk, (int)head_dim, scale, bucket, bias_tab,
                       (int)heads, key_mask, key_mask_bs);
    return scail_check_launch("attn_small");
}

extern "C" int scail_mul_bf16(const scail_bf16* a, const scail_bf16* b, scail_bf16* y, int64_t n, void* stream) {
    SCAIL_REQUIRE(n % 8 == 0, "n must be a multiple of 8");
    if (n == 0) return 0;
    hipLaunchKernelGGL(mul_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, n / 8);
    return scail_check_launch("mul_bf16");
}

extern "C" int scail_row_affine(const scail_bf16* x, scail_bf16* y, const float* rowscale, const scail_bf16* addrow, int64_t add_rows,
                                int64_t rows, int64_t D, void* stream) {
    SCAIL_REQUIRE(D % 8 == 0, "D must be a multiple of 8");
    if (rows == 0) return 0;
    const int64_t total = rows * (D / 8);
    hipLaunchKernelGGL(row_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, rowscale,
                       addrow, add_rows > 0 ? add_rows : 1, rows, (int)(D / 8));
    return scail_check_launch("row_affine");
}

// ------------------------------------------------------------------------------------------------
// MEASUREMENT helper: a collective's CU occupancy on one GPU (include/scail_hip.h scail_comm_standin)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void comm_standin_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16, int64_t min_ticks) {
    const int64_t t0 = (int64_t)wall_clock64();                    // 100 MHz constant clock
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
    while ((int64_t)wall_clock64() - t0 < min_ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int scail_comm_standin(const void* src, void* dst, int64_t bytes, int32_t workgroups, int64_t min_ns, void* stream) {
    SCAIL_REQUIRE(bytes >= 0 && bytes % 16 == 0 && aligned16(src) && aligned16(dst), "bytes must be a multiple of 16, pointers 16-byte aligned");
    SCAIL_REQUIRE(workgroups >= 1 && workgroups <= 4096 && min_ns >= 0 && min_ns <= 1000000000ll, "workgroups in [1, 4096], min_ns in [0, 1e9]");
    hipLaunchKernelGGL(comm_standin_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), bytes / 16, min_ns / 10);
    return scail_check_launch("comm_standin");
}

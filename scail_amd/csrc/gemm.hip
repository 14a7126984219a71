// bf16 GEMM  y = epilogue(x . W^T + bias)  on MFMA 32x32x16 (gfx950), fp32 accumulate.
//
// Tile 128(m) x 128(n) x 64(k), 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA fragments.
// Both operands are K-contiguous in HBM (x rows, torch-Linear W rows), so both tiles are staged
// row-major into LDS through registers (16-byte global loads, 16-byte ds_write), rows padded to
// 144 B so that the 16-lane service groups of ds_read_b128 hit 16 distinct 16-byte slots (no bank
// conflicts), double-buffered with ONE barrier per k-tile: the global loads of tile t+1 are issued
// before the MFMAs of tile t and written to the other buffer after them.
//
// The MFMA is issued "transposed" (A operand = W fragment, B operand = x fragment), so a lane's
// accumulator registers run along n: 4 consecutive output columns per register quad -> 8-byte
// stores and float4 bias / gate loads in the epilogue.
//
// Block -> tile map: XCD-aware remap (block b runs on XCD b % 8; give each XCD a contiguous range
// of tiles so neighbours share operand panels in that XCD's L2), then grouped ordering (g_group_m = 4 m-tiles
// per group, m fastest) so the co-resident tiles of an XCD share a few operand panels.
#include "common.h"
#include <atomic>
#include <algorithm>
#include <map>
#include <tuple>
#include <mutex>
#include <vector>

#define BK 64
#define LDT 72  // padded LDS row, elements (144 B)
#define GROUP_M 8

struct GemmParams {
    const u16* x; int64_t lda;
    const u16* w;
    const float* bias;
    u16* y; int64_t ldc;
    int M, N, K;
    const u16* resid; int64_t ldr;
    const float* gate; int64_t gate_stride; int64_t rows_per_batch;
    int group_m;   // m-tiles per tile group of the block -> tile map (q8 kernel; others use GROUP_M)
};

// Two instantiations share this body:
//   128 x 128 tile, 4 waves (2 x 2), wave tile 64 x 64,  74 KB LDS, 2 workgroups per CU  (small / ragged N)
//   256 x 256 tile, 8 waves (2 x 4), wave tile 128 x 64, 147 KB LDS, 1 workgroup per CU  (the big per-token
//   GEMMs: half the L2->LDS bytes per FLOP, 6 LDS fragment reads per 8 MFMAs instead of 4 per 4)
// Fused epilogue shared by the kernels below.  A lane holds output column m (= row of x) and, per 32x32
// accumulator fragment, rows n = nbase + 8 rr + 4 g + e.  All loads of one pass (bias once; residual and
// gate per output row) are issued before their first use: one memory round trip per pass.
template <int EPI, int MI, int NI, int WTM, int WTN>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[NI][MI], const GemmParams& p, int m0, int n0, int wm, int wn,
                                              int l31, int g) {
    float4 bb[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int n = n0 + wn * WTN + ni * 32 + 8 * rr + 4 * g;
            bb[ni][rr] = (p.bias != nullptr && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * WTM + mi * 32 + l31;
        if (m >= p.M) continue;
        uint2 rv[NI][4];
        float4 gt[NI][4];
        if (EPI == SCAIL_EPI_RESID) {
            const int64_t bidx = (p.gate != nullptr) ? (int64_t)m / p.rows_per_batch : 0;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int n = n0 + wn * WTN + ni * 32 + 8 * rr + 4 * g;
                    const bool ok = n < p.N;
                    rv[ni][rr] = ok ? *reinterpret_cast<const uint2*>(p.resid + (int64_t)m * p.ldr + n) : make_uint2(0, 0);
                    gt[ni][rr] = (ok && p.gate != nullptr) ? *reinterpret_cast<const float4*>(p.gate + bidx * p.gate_stride + n)
                                                           : make_float4(1.f, 1.f, 1.f, 1.f);
                }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int n = n0 + wn * WTN + ni * 32 + 8 * rr + 4 * g;
                if (n >= p.N) continue;
                float v[4];
                v[0] = acc[ni][mi][4 * rr + 0] + bb[ni][rr].x;
                v[1] = acc[ni][mi][4 * rr + 1] + bb[ni][rr].y;
                v[2] = acc[ni][mi][4 * rr + 2] + bb[ni][rr].z;
                v[3] = acc[ni][mi][4 * rr + 3] + bb[ni][rr].w;
                if (EPI == SCAIL_EPI_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
                } else if (EPI == SCAIL_EPI_GELU_ERF) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                } else if (EPI == SCAIL_EPI_RESID) {
                    v[0] = bf_lo(rv[ni][rr].x) + gt[ni][rr].x * v[0];
                    v[1] = bf_hi(rv[ni][rr].x) + gt[ni][rr].y * v[1];
                    v[2] = bf_lo(rv[ni][rr].y) + gt[ni][rr].z * v[2];
                    v[3] = bf_hi(rv[ni][rr].y) + gt[ni][rr].w * v[3];
                }
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(p.y + (int64_t)m * p.ldc + n) = o;
            }
        }
    }
}

// DMA: operand tiles arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 rows x 128 B per wave-instruction,
// lane-linear -> unpadded 128-B rows); bank conflicts are removed by XOR-ing the 16-byte chunk index with
// (row >> 1) & 7 on the per-lane SOURCE address and on the fragment reads.  No staging VGPRs, no ds_write.
// ABL (ablation, wrong results, tools/microbench.py only): bit 0 = no operand loads inside the k loop,
// bit 1 = no barrier inside the k loop.
template <int BM, int BN, int WM, int WN, int EPI, bool DMA, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_kernel(GemmParams p) {
    constexpr int NT = 64 * WM * WN;      // threads
    constexpr int LDX = DMA ? BK : LDT;   // LDS row length (elements)
    constexpr int RS = NT / 8;            // tile rows covered by one staging pass (8 x 16 B chunks per row)
    static_assert(DMA || (BM / RS == 4 && BN / RS == 4), "the register staging code below is written for 4 passes per operand");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* Xs = smem;                  // [2][BM][LDX]
    u16* Ws = smem + 2 * BM * LDX;   // [2][BN][LDX]

    // ---- tile mapping ----
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nb = tiles_m * tiles_n;
    int wg;
    {
        const int id = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, loc = id >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int in_group = GROUP_M * tiles_n;
    const int gid = wg / in_group;
    const int first_m = gid * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int pid_m = first_m + (wg % in_group) % gsz;
    const int pid_n = (wg % in_group) / gsz;
    const int m0 = pid_m * BM, n0 = pid_n * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, g = lane >> 5;

    // ---- staging: chunk = tid + NT i  -> row = (tid >> 3) + RS i, kc = tid & 7 ----
    const int srow = tid >> 3, kc = tid & 7;
    // (named scalars, not arrays: hipcc sends register arrays written under a branch to scratch)
#define ROWPTR(i_)                                                                              \
    const u16* xptr##i_ = p.x + (int64_t)min(m0 + srow + RS * i_, p.M - 1) * p.lda + kc * 8;    \
    const u16* wptr##i_ = p.w + (int64_t)min(n0 + srow + RS * i_, p.N - 1) * p.K + kc * 8;      \
    uint4 xr##i_, wr##i_;
    ROWPTR(0) ROWPTR(1) ROWPTR(2) ROWPTR(3)
#define G_LOAD1(i_, k0_)                                              \
    xr##i_ = *reinterpret_cast<const uint4*>(xptr##i_ + (k0_));       \
    wr##i_ = *reinterpret_cast<const uint4*>(wptr##i_ + (k0_));
#define G_LOAD(k0_) G_LOAD1(0, k0_) G_LOAD1(1, k0_) G_LOAD1(2, k0_) G_LOAD1(3, k0_)
#define S_STORE1(i_, buf_)                                                                        \
    *reinterpret_cast<uint4*>(Xs + ((buf_) * BM + srow + RS * i_) * LDT + kc * 8) = xr##i_;       \
    *reinterpret_cast<uint4*>(Ws + ((buf_) * BN + srow + RS * i_) * LDT + kc * 8) = wr##i_;
#define S_STORE(buf_) S_STORE1(0, buf_) S_STORE1(1, buf_) S_STORE1(2, buf_) S_STORE1(3, buf_)

    // LDS-DMA pieces: piece j of an operand = tile rows 8j..8j+7; lane -> row 8j + (l >> 3), chunk l & 7
    constexpr int PA = BM / 8 / (NT / 64), PB = BN / 8 / (NT / 64);   // pieces per wave
    const int d_row = lane >> 3, d_c = lane & 7;
    const int lm0 = (ABL & 8) ? 0 : m0, ln0 = (ABL & 8) ? 0 : n0;   // ablation: every block loads tile (0,0): all L2 hits
#define DMA_ISSUE(k0_, buf_)                                                                                  \
    {                                                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < PA; ++i_) {                                                   \
            const int r_ = 8 * (wave * PA + i_) + d_row;                                                      \
            const u16* src_ = p.x + (int64_t)min(lm0 + r_, p.M - 1) * p.lda + ((ABL & 16) ? 0 : (k0_)) + ((d_c ^ ((r_ >> 1) & 7)) << 3);  \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,              \
                (__attribute__((address_space(3))) void*)(Xs + ((buf_) * BM + 8 * (wave * PA + i_)) * LDX), 16, 0, 0); \
        }                                                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < PB; ++i_) {                                                   \
            const int r_ = 8 * (wave * PB + i_) + d_row;                                                      \
            const u16* src_ = p.w + (int64_t)min(ln0 + r_, p.N - 1) * p.K + ((ABL & 16) ? 0 : (k0_)) + ((d_c ^ ((r_ >> 1) & 7)) << 3);    \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,              \
                (__attribute__((address_space(3))) void*)(Ws + ((buf_) * BN + 8 * (wave * PB + i_)) * LDX), 16, 0, 0); \
        }                                                                                                     \
    }
    // one A piece + one B piece (index part_) of this wave: spreading the 2 x PA issues over the k-steps keeps
    // the wave's MFMA stream from stalling behind 8 back-to-back DMA issues (each costs ~100 issue cycles)
#define DMA_ISSUE_PART(k0_, buf_, part_)                                                                      \
    {                                                                                                         \
        if ((part_) < PA) {                                                                                   \
            const int r_ = 8 * (wave * PA + (part_)) + d_row;                                                 \
            const u16* src_ = p.x + (int64_t)min(m0 + r_, p.M - 1) * p.lda + (k0_) + ((d_c ^ ((r_ >> 1) & 7)) << 3);  \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,              \
                (__attribute__((address_space(3))) void*)(Xs + ((buf_) * BM + 8 * (wave * PA + (part_))) * LDX), 16, 0, 0); \
        }                                                                                                     \
        if ((part_) < PB) {                                                                                   \
            const int r_ = 8 * (wave * PB + (part_)) + d_row;                                                 \
            const u16* src_ = p.w + (int64_t)min(n0 + r_, p.N - 1) * p.K + (k0_) + ((d_c ^ ((r_ >> 1) & 7)) << 3);    \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,              \
                (__attribute__((address_space(3))) void*)(Ws + ((buf_) * BN + 8 * (wave * PB + (part_))) * LDX), 16, 0, 0); \
        }                                                                                                     \
    }
    int foff[BK / 16];   // fragment-read offset (elements) of chunk (2 ks + g) in this lane's row
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) foff[ks] = DMA ? (((2 * ks + g) ^ ((l31 >> 1) & 7)) << 3) : (ks * 16 + g * 8);

    f32x16 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int nk = p.K / BK;
    if (DMA) {
        DMA_ISSUE(0, 0)
    } else {
        G_LOAD(0)
        S_STORE(0)
    }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk && !(ABL & 1)) {
            if (DMA) {
                if (!(ABL & 4)) DMA_ISSUE((t + 1) * BK, cur ^ 1)
            } else {
                G_LOAD((t + 1) * BK)
            }
        }
        const u16* xs = Xs + (cur * BM + wm * WTM + l31) * LDX;
        const u16* ws = Ws + (cur * BN + wn * WTN + l31) * LDX;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 wf[NI], xf[MI];
#pragma unroll
            for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(ws + i * 32 * LDX + foff[ks]);
#pragma unroll
            for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xs + i * 32 * LDX + foff[ks]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
            if (DMA && (ABL & 4) && t + 1 < nk) DMA_ISSUE_PART((t + 1) * BK, cur ^ 1, ks)
        }
        if (!DMA) { if (t + 1 < nk) { S_STORE(cur ^ 1) } }
        if (!(ABL & 2)) __syncthreads();
    }

    gemm_epilogue<EPI, MI, NI, WTM, WTN>(acc, p, m0, n0, wm, wn, l31, g);
}

// ------------------------------------------------------------------------------------------------
// Removed after measurement (in the history; numbers in DESIGN.md section 4.1): a 4-deep ring of HALF k-tiles with
// counted vmcnt (1025 TFLOP/s) and a ping-pong of the two wave rows on half k-tiles with raised MFMA priority (1000),
// both below the lock-step LDS-DMA kernel (1090) and the quadrant-phase kernel below (1250-1340).
// ------------------------------------------------------------------------------------------------

// q8 tile-group height (m-tiles sharing an n sweep per group): 4 measured best on the four per-token GEMM shapes
// (tools/gemm_group_probe.py: 1: -5 %, 2: -1 %, 3-4: best, 8: -1..2 %, 16: -8 %, 32: -18 %)
static int g_group_m = 4;
#ifdef SCAIL_ABLATIONS
// ---- measurement build only (include/scail_hip_ablation.h) ----
int scail_gemm_group_m(int v) { g_group_m = v > 0 ? v : 4; return 0; }
static int g_gemm_tile = 0;  // 0: choose by shape; 128 / 256: force; 257: 256 tile + LDS-DMA; 260: + DMA spread over the k-steps; 261: q8; 262: q8, MFMA-wave priority; 266: 4 waves
// Codes >= 1000 select TIMING ABLATIONS of the big-tile kernels (most of them compute wrong results on purpose: loads or
// fragment reads removed, all-L2-hit addressing, ...).  They exist for tools/microbench.py.
int scail_gemm_tune(int v) {
    bool ok = (v >= 1000 && v < 1600);
    switch (v) {
        case 0: case 128: case 256: case 257: case 260: case 261: case 262: case 266: ok = true; break;
        default: break;
    }
    if (!ok) {
        scail_set_error("scail_tune_set: gemm_tile " + std::to_string(v) + " is not a known tile code");
        return 1;
    }
    g_gemm_tile = v;
    return 0;
}
#endif

// Summed lifetime of the q8 workgroups in s_memtime ticks and their count; written only by the ABL-bit-128 instantiations of
// the measurement build (tools/microbench.py: sum / 256 CUs / wall time = the clock the chip sustains under a variant).
__device__ unsigned long long g_q8_clk[2] = {0ull, 0ull};
#ifdef SCAIL_ABLATIONS
int scail_attn_clk(unsigned long long* out2, int reset);   // attn.hip: same counters for the attention kernel
extern "C" int scail_debug_cycles(unsigned long long* out2, int reset) {
    if (out2 != nullptr && hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_q8_clk), 16) != hipSuccess) {
        scail_set_error("scail_debug_cycles: hipMemcpyFromSymbol failed");
        return 2;
    }
    if (scail_attn_clk(out2, reset) != 0) {
        scail_set_error("scail_debug_cycles: attention counters");
        return 2;
    }
    if (reset) {
        const unsigned long long init[2] = {0ull, 0ull};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_q8_clk), init, 16) != hipSuccess) {
            scail_set_error("scail_debug_cycles: hipMemcpyToSymbol failed");
            return 2;
        }
    }
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------------
// Quadrant-phase kernel ("q8", gemm_tile 261).  256 x 256 x 64 tile, 8 waves (2 x 4), wave tile 128(m) x 64(n)
// worked as FOUR 64 x 32 quadrants per k-tile, each = 8 MFMA 32x32x16 (256 matrix-pipe cycles):
//     phase 0: read a0 block 1 (4 x b128) + b0 (4)       MFMA a0.b0     a0/a1 = rows 0-63 / 64-127 of the wave's 128
//     phase 1: read b1 (4)                               MFMA a0.b1     b0/b1 = cols 0-31 / 32-63 of the wave's 64
//     phase 2: read a1 (8)                               MFMA a1.b1     (block = 32 rows)
//     phase 3: read a0 block 0 of the NEXT k-tile (4)    MFMA a1.b0
// The two wave rows (waves 0-3 / 4-7; wave w and w+4 share a SIMD) run one barrier interval apart, so in every
// interval one wave per SIMD is in its MFMA section while its partner reads fragments and issues LDS-DMA.
// Operands are staged as four 16 KB "streams" per k-tile, each the rows one phase reads
//     Af = a0 rows of both wave rows, As = a1 rows, Bf = b0 rows of the four wave columns, Bs = b1 rows
// (2 DMA pieces per wave per stream: buffer_load_dwordx4 ... lds, tile origin in the descriptor, 32-bit lane
// offsets), double-buffered per stream.  One stream is issued per phase:
//     phase 0 of tile T: Bs(T+1)   phase 1: As(T+1)   phase 2: Af(T+2)   phase 3: Bf(T+2)
// WAR: each is issued >= 2 phases after the last read of the slot it overwrites (the reads were retired by an
// lgkmcnt(0) and a barrier lies between).  RAW: every phase ends its read section with a counted s_waitcnt: vmcnt(8) (4
// streams in flight) is the loosest legal count -- the stream issued in phase g-4 is then retired by every wave in phase g
// and first read in phase g+1 or later, at least one barrier after the slowest wave's wait; any smaller count only waits
// more (measured: vmcnt(6), (4) and (2) are all within noise of (8) -- the loads land well within two phases).
// Measured (M = 97 664, N = 15 360, K = 5120): 1.21 PFLOP/s vs 1.09 for the lock-step DMA kernel; ablations:
// no DMA 1.50, no fragment reads 1.30, neither 1.64 (barrier-paced MFMA only), all-L2-hit operands 1.27,
// no vmcnt waits +1 %, 4-byte instead of 16-byte DMA pieces +-0 (the DMA cost is per instruction, ~80 issue
// cycles, not per byte).  global_load_lds instead of the buffer form: -5 %; DMA issued inside the MFMA section:
// -4 %; DMA before the fragment reads: -1.5 %.
// ------------------------------------------------------------------------------------------------
template <int EPI, int ABL = 0>   // ABL (microbench only, wrong results): 1 no DMA in the loop, 2 no fragment reads
                                  // in the loop, 4 no vmcnt waits, 8 every block loads tile (0,0)
__global__ __launch_bounds__(512) void gemm_bf16_q8_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 256, WN = 4;
    constexpr int WTM = 128, WTN = 64, MI = 4, NI = 2;
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* Xs = smem;                    // [2][BM][BK]
    u16* Ws = smem + 2 * BM * BK;      // [2][BN][BK]

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nb = tiles_m * tiles_n;
    int wg;
    {
        const int id = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, loc = id >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int in_group = p.group_m * tiles_n;
    const int gid = wg / in_group;
    const int first_m = gid * p.group_m;
    const int gsz = min(tiles_m - first_m, p.group_m);
    const int pid_m = first_m + (wg % in_group) % gsz;
    const int pid_n = (wg % in_group) / gsz;
    const int m0 = pid_m * BM, n0 = pid_n * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, g = lane >> 5;
    const unsigned long long clk0 = (ABL & 128) ? __builtin_amdgcn_s_memtime() : 0ull;

    // DMA: piece j = 2 wave + i of a stream = its rows r' = 8j .. 8j+7 (lane -> row 8j + (l >> 3), chunk l & 7);
    // stream row r' -> tile row:  A streams (r' >> 6) * 128 + (r' & 63) [+ 64 for As],  B (r' >> 5) * 64 + (r' & 31) [+ 32]
    // lane byte offsets from the tile origin (rows past M / N clamp to the last row; their outputs are not stored)
    const int d_row = lane >> 3, d_c = lane & 7;
    const u16* xt = p.x + (int64_t)((ABL & 8) ? 0 : m0) * p.lda;     // tile origins (uniform)
    const u16* wt = p.w + (int64_t)((ABL & 8) ? 0 : n0) * p.K;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(xt), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(wt), 0, 0x7FFFFFFF, 0x00020000);
#define Q8_ROWS(i_)                                                                                         \
    const int ja##i_ = 8 * (2 * wave + i_);                                                                 \
    const int rbAf##i_ = (ja##i_ >> 6) * 128 + (ja##i_ & 63), rbBf##i_ = (ja##i_ >> 5) * 64 + (ja##i_ & 31);  \
    const int oAf##i_ = 2 * ((min(m0 + rbAf##i_ + d_row, p.M - 1) - m0) * (int)p.lda + ((d_c ^ (((rbAf##i_ + d_row) >> 1) & 7)) << 3));            \
    const int oAs##i_ = 2 * ((min(m0 + rbAf##i_ + 64 + d_row, p.M - 1) - m0) * (int)p.lda + ((d_c ^ (((rbAf##i_ + 64 + d_row) >> 1) & 7)) << 3));  \
    const int oBf##i_ = 2 * ((min(n0 + rbBf##i_ + d_row, p.N - 1) - n0) * p.K + ((d_c ^ (((rbBf##i_ + d_row) >> 1) & 7)) << 3));                   \
    const int oBs##i_ = 2 * ((min(n0 + rbBf##i_ + 32 + d_row, p.N - 1) - n0) * p.K + ((d_c ^ (((rbBf##i_ + 32 + d_row) >> 1) & 7)) << 3));
    Q8_ROWS(0) Q8_ROWS(1)
    // ABL 256 (timing only): every DMA piece reads 1 KB of CONTIGUOUS memory (as if the operands were stored
    // tile-major) instead of 8 rows x 128 B, 512: only the W pieces do
#define Q8_CONTIG(o_, j_, str_) ((ABL & (str_)) ? (int)(lane * 16 + (j_) * 1024) : (o_))
    const int cAf0 = Q8_CONTIG(oAf0, 2 * wave, 256), cAf1 = Q8_CONTIG(oAf1, 2 * wave + 1, 256);
    const int cAs0 = Q8_CONTIG(oAs0, 16 + 2 * wave, 256), cAs1 = Q8_CONTIG(oAs1, 17 + 2 * wave, 256);
    const int cBf0 = Q8_CONTIG(oBf0, 2 * wave, 256 | 512), cBf1 = Q8_CONTIG(oBf1, 2 * wave + 1, 256 | 512);
    const int cBs0 = Q8_CONTIG(oBs0, 16 + 2 * wave, 256 | 512), cBs1 = Q8_CONTIG(oBs1, 17 + 2 * wave, 256 | 512);
    typedef unsigned int q8_u32x4 __attribute__((ext_vector_type(4)));
    q8_u32x4 dmy0 = {0u, 0u, 0u, 0u};      // ABL 64: loads to (dead) VGPRs instead of LDS-DMA, to price the DMA issue itself
#define Q8_DMA(rs_, voff_, soff_, dst_)                                                             \
    if (ABL & 64) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dmy0) : "v"(voff_), "s"(rs_), "s"(soff_));  \
    else if (ABL & 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (__attribute__((address_space(3))) void*)(dst_), 4, voff_, soff_, 0, 0);  \
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (__attribute__((address_space(3))) void*)(dst_), 16, voff_, soff_, 0, 0);
#define Q8_ISSUE_A(tile_, radd_, o0_, o1_)                                                          \
    {                                                                                               \
        u16* d_ = Xs + ((tile_) & 1) * BM * BK;                                                     \
        Q8_DMA(rsA, o0_, (tile_) * (BK * 2), d_ + (rbAf0 + radd_) * BK)                             \
        Q8_DMA(rsA, o1_, (tile_) * (BK * 2), d_ + (rbAf1 + radd_) * BK)                             \
    }
#define Q8_ISSUE_B(tile_, radd_, o0_, o1_)                                                          \
    {                                                                                               \
        u16* d_ = Ws + ((tile_) & 1) * BN * BK;                                                     \
        Q8_DMA(rsB, o0_, (tile_) * (BK * 2), d_ + (rbBf0 + radd_) * BK)                             \
        Q8_DMA(rsB, o1_, (tile_) * (BK * 2), d_ + (rbBf1 + radd_) * BK)                             \
    }
#define Q8_VM8 __builtin_amdgcn_s_waitcnt(0x0F78);    // vmcnt(8)
#define Q8_VM0 __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0)
#define Q8_LGKM0 __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#define Q8_SB __builtin_amdgcn_sched_barrier(0);

    // fragment reads: lane -> row l31 of a 32-row block, 16-byte chunk (2 ks + g) ^ ((row >> 1) & 7)
    const int e0 = ((g ^ ((l31 >> 1) & 7)) << 3);
    const u16* xs0 = Xs + (wm * WTM + l31) * BK;
    const u16* ws0 = Ws + (wn * WTN + l31) * BK;

    f32x16 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    bf16x8 fa[2][4], fn[4], fb0[4], fb1[4];     // fn = block 0 of a0, read one phase ahead of its tile
#define Q8_READ_BLK(dst_, buf_, blk_)                                                               \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                             \
        dst_[ks_] = *reinterpret_cast<const bf16x8*>(xs0 + ((buf_) * BM + (blk_) * 32) * BK + (e0 ^ (ks_ << 4)));
#define Q8_READ_B(dst_, buf_, half_)                                                                \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                             \
        dst_[ks_] = *reinterpret_cast<const bf16x8*>(ws0 + ((buf_) * BN + (half_) * 32) * BK + (e0 ^ (ks_ << 4)));
    // Issue priority goes to the wave in its read/DMA section (s_setprio 1 outside the MFMA section): the MFMA wave
    // needs one issue slot per 32 cycles, the loader is issue-bound (+2 % measured; ABL bit 32 = the other way round).
    // MFMAs are pure register ops: nothing orders them against s_barrier / s_setprio for the compiler, and hipcc does
    // sink them into the next phase's read section.  The empty volatile asms tie the operands (after the barrier)
    // and the accumulators (before the next barrier) to the instruction stream.
#define Q8_PIN4(f_) asm volatile("" : "+v"(f_[0]), "+v"(f_[1]), "+v"(f_[2]), "+v"(f_[3]));
#define Q8_MFMA(ni_, mi0_, fa0_, fa1_, fb_)                                                         \
    Q8_PIN4(fa0_) Q8_PIN4(fa1_) Q8_PIN4(fb_)                                                        \
    __builtin_amdgcn_s_setprio((ABL & 32) ? 1 : 0);                                                 \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) {                                           \
        acc[ni_][mi0_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb_[ks_], fa0_[ks_], acc[ni_][mi0_], 0, 0, 0);             \
        acc[ni_][mi0_ + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb_[ks_], fa1_[ks_], acc[ni_][mi0_ + 1], 0, 0, 0);     \
    }                                                                                               \
    __builtin_amdgcn_s_setprio((ABL & 32) ? 0 : 1);                                                 \
    asm volatile("" : "+v"(acc[ni_][mi0_]), "+v"(acc[ni_][mi0_ + 1]));
    // one phase: fragment reads, this phase's stream issue + counted wait, barrier, MFMA quadrant, barrier
#define Q8_PHASE(READS_, COND_, ISSUE_, ni_, mi0_, fa0_, fa1_, fb_)                                 \
    if (!(ABL & 2) || t == 0) { READS_ }                                                            \
    if (!(ABL & 1) && (COND_)) { ISSUE_ if (!(ABL & 4)) { Q8_VM8 } } else { Q8_VM0 }  \
    Q8_SB __builtin_amdgcn_s_barrier(); Q8_LGKM0 Q8_SB                                              \
    Q8_MFMA(ni_, mi0_, fa0_, fa1_, fb_)                                                             \
    Q8_SB __builtin_amdgcn_s_barrier(); Q8_SB
#define Q8_TILE(C1_, C2_)                                                                           \
    {                                                                                               \
        const int buf = t & 1;                                                                      \
        Q8_PHASE(Q8_READ_BLK(fa[1], buf, 1) Q8_READ_B(fb0, buf, 0), C1_, Q8_ISSUE_B(t + 1, 32, cBs0, cBs1), 0, 0, fn, fa[1], fb0)  \
        Q8_PHASE(Q8_READ_B(fb1, buf, 1), C1_, Q8_ISSUE_A(t + 1, 64, cAs0, cAs1), 1, 0, fn, fa[1], fb1)                            \
        Q8_PHASE(Q8_READ_BLK(fa[0], buf, 2) Q8_READ_BLK(fa[1], buf, 3), C2_, Q8_ISSUE_A(t + 2, 0, cAf0, cAf1), 1, 2, fa[0], fa[1], fb1) \
        Q8_PHASE(Q8_READ_BLK(fn, buf ^ 1, 0), C2_, Q8_ISSUE_B(t + 2, 0, cBf0, cBf1), 0, 2, fa[0], fa[1], fb0)                      \
    }

    const int nk = p.K / BK;
    // prologue: all four streams of tile 0, Af and Bf of tile 1
    Q8_ISSUE_A(0, 0, cAf0, cAf1)
    Q8_ISSUE_B(0, 0, cBf0, cBf1)
    Q8_ISSUE_B(0, 32, cBs0, cBs1)
    Q8_ISSUE_A(0, 64, cAs0, cAs1)
    if (nk > 1) {
        Q8_ISSUE_A(1, 0, cAf0, cAf1)
        Q8_ISSUE_B(1, 0, cBf0, cBf1)
        Q8_VM8                      // Af(0), Bf(0) landed (this wave's pieces)
    } else {
        Q8_VM0
    }
    __builtin_amdgcn_s_barrier();
    Q8_READ_BLK(fn, 0, 0)
    if (wm == 1) __builtin_amdgcn_s_barrier();      // wave row 1 runs one interval behind row 0

    int t = 0;
    for (; t + 2 < nk; ++t) Q8_TILE(true, true)
    for (; t < nk; ++t) Q8_TILE(t + 1 < nk, false)
    if (wm == 0) __builtin_amdgcn_s_barrier();      // re-align the two wave rows
    if (ABL & 64) { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" :: "v"(dmy0)); }
    if ((ABL & 128) && tid == 0) {
        atomicAdd(&g_q8_clk[0], (unsigned long long)__builtin_amdgcn_s_memtime() - clk0);
        atomicAdd(&g_q8_clk[1], 1ull);
    }
    gemm_epilogue<EPI, MI, NI, WTM, WTN>(acc, p, m0, n0, wm, wn, l31, g);
}

template <int BM, int BN, int WM, int WN, int EPI, bool DMA, int ABL = 0>
static int launch_gemm_t(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = 2 * (BM + BN) * (DMA ? BK : LDT) * 2;
    static ScailDeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, WM, WN, EPI, DMA, ABL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            scail_set_error(std::string("gemm: hipFuncSetAttribute failed: ") + hipGetErrorString(e));
            return 2;
        }
        attr_set.done();
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, EPI, DMA, ABL>), dim3((unsigned)tiles), dim3(64 * WM * WN), lds, stream, p);
    return scail_check_launch("gemm_bf16");
}

template <int EPI, int ABL = 0>
static int launch_gemm_q8(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = 2 * (256 + 256) * BK * 2;   // 128 KB
    static ScailDeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_q8_kernel<EPI, ABL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            scail_set_error(std::string("scail_gemm_bf16: hipFuncSetAttribute: ") + hipGetErrorString(e));
            return 2;
        }
        attr_set.done();
    }
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    hipLaunchKernelGGL((gemm_bf16_q8_kernel<EPI, ABL>), dim3((unsigned)tiles), dim3(512), lds, stream, p);
    return scail_check_launch("gemm_bf16");
}

template <int EPI>
static int launch_gemm(const GemmParams& p, hipStream_t stream) {
#ifdef SCAIL_ABLATIONS
    // forced tiles / schedules and timing ablations of the measurement build (scail_gemm_tune)
    if (g_gemm_tile == 261) return launch_gemm_q8<EPI>(p, stream);
    if (g_gemm_tile == 266) return launch_gemm_t<256, 256, 2, 2, EPI, true>(p, stream);   // 4 waves, 128 x 128 per wave (AGPR accumulators)
    if (EPI == 0 && g_gemm_tile == 1101) return launch_gemm_q8<0, 1>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1102) return launch_gemm_q8<0, 2>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1103) return launch_gemm_q8<0, 3>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1104) return launch_gemm_q8<0, 4>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1108) return launch_gemm_q8<0, 8>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1112) return launch_gemm_q8<0, 12>(p, stream);
    if (g_gemm_tile == 262) return launch_gemm_q8<EPI, 32>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1300) return launch_gemm_q8<0, 128>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1301) return launch_gemm_q8<0, 129>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1302) return launch_gemm_q8<0, 130>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1303) return launch_gemm_q8<0, 131>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1364) return launch_gemm_q8<0, 192>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1256) return launch_gemm_q8<0, 256>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1512) return launch_gemm_q8<0, 512>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1164) return launch_gemm_q8<0, 64>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1116) return launch_gemm_q8<0, 16>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1124) return launch_gemm_q8<0, 24>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1001) return launch_gemm_t<256, 256, 2, 4, 0, true, 1>(p, stream);   // ablations
    if (EPI == 0 && g_gemm_tile == 1002) return launch_gemm_t<256, 256, 2, 4, 0, true, 2>(p, stream);
    if (g_gemm_tile == 260) return launch_gemm_t<256, 256, 2, 4, EPI, true, 4>(p, stream);   // DMA issue spread over the k-steps
    if (EPI == 0 && g_gemm_tile == 1008) return launch_gemm_t<256, 256, 2, 4, EPI, true, 8>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1024) return launch_gemm_t<256, 256, 2, 4, EPI, true, 24>(p, stream);
    if (EPI == 0 && g_gemm_tile == 1003) return launch_gemm_t<256, 256, 2, 4, 0, true, 3>(p, stream);
    // measured at M = 97 664 (profiles/r01_pmc.md): 128 tile 820, 256 tile 1000, 256 + LDS-DMA 1090, 256 + DMA ring
    // of half k-tiles with counted vmcnt 1025, ping-pong 1000, quadrant-phase q8 1240-1290 TFLOP/s (default for the
    // big per-token GEMMs; the vendor library's assembly kernel reaches 1500 on the same shapes)
    if (g_gemm_tile == 256) return launch_gemm_t<256, 256, 2, 4, EPI, false>(p, stream);
    if (g_gemm_tile == 257) return launch_gemm_t<256, 256, 2, 4, EPI, true>(p, stream);
    if (g_gemm_tile == 128) return launch_gemm_t<128, 128, 2, 2, EPI, false>(p, stream);
#endif
    const bool big = p.M >= 2048 && p.N >= 1024;
    // q8 addresses a tile through a 2 GB buffer descriptor with 32-bit lane offsets
    if (big && 512 * p.lda + 2 * (int64_t)p.K < (1ll << 31) && 514 * (int64_t)p.K < (1ll << 31)) return launch_gemm_q8<EPI>(p, stream);
    if (big) return launch_gemm_t<256, 256, 2, 4, EPI, true>(p, stream);
    return launch_gemm_t<128, 128, 2, 2, EPI, false>(p, stream);
}

// ================================================================================================
// gemm4: the hand-scheduled 4-wave kernels (256 x 256 x 64 tile, one wave per SIMD, wave tile 128 x 128 as 8 x 8 blocks of
// v_mfma_f32_16x16x32_bf16, the 256 accumulators of a lane in a[0:255], LDS-DMA staging two tiles deep inside two 64 KB slots, the
// fragments of a whole k-tile in registers and read half a tile ahead).  gfx950 assembly GENERATED by scail_amd/asmgen/gemm4.py
// (csrc/gemm4.s), embedded as a code object and loaded with hipModuleLoadData on first use.  Kernel argument block =
// asmgen/gemm4.py KERNARG_FMT.  Default for the big per-token GEMMs (1467 TFLOP/s where the hipcc q8
// kernel does 1309 and the vendor's assembly kernel 1495).
// ================================================================================================
static const unsigned char k_gemm4_hsaco[] = {
#include "gemm4_hsaco.inc"
};
#ifdef SCAIL_ABLATIONS
// gemm8 (measurement build): two waves per SIMD (8 waves, wave tile 128 x 64, 128 accumulators in a[0:127]; asmgen/gemm8.py)
static const unsigned char k_gemm8_hsaco[] = {
#include "gemm8_hsaco.inc"
};
#endif
struct Gemm4Args {
    const void* x; const void* w; const void* bias; void* y; const void* resid; const void* gate; const void* table;
    int64_t lda, ldc, ldr, gs;
    int32_t M, N, K, rows_per_batch;
    int32_t grid, entries;      // persistent kernels: workgroups launched (a multiple of 8) and entries of the order table
};
static_assert(sizeof(Gemm4Args) == 112, "Gemm4Args must match asmgen/gemm4.py KERNARG_SIZE");
// per-device caches (a code object / a hipMalloc'd table belongs to the device that was current when it was created)
static std::map<std::pair<int, int>, hipModule_t> g_gemm4_modules;          // (device, 4 | 8) -> loaded code object
static std::map<std::pair<int, std::string>, hipFunction_t> g_gemm4_fn;     // (device, kernel name)
static std::map<std::tuple<int, int, int, int, int>, std::pair<uint32_t*, int>> g_gemm4_tables;   // (device, m tiles, n tiles, group height, table mode) -> device order table, entries
static std::mutex g_gemm4_mutex;
static std::atomic<int> g_gemm4_mode{4};               // generated kernels where eligible: 4 = gemm4 (default), 0 = never (csrc/gemm.hip only), 8 = gemm8 (measurement build)
static std::string g_gemm4_suffix;         // A/B variants of the measurement build ("gemm4_kernel:<suffix>")
static int g_gemm4_table_mode = 1;         // tile -> XCD assignment of the order table (0, 2: measurement build A/B, knob "gemm4_table")

static int gemm4_function(const std::string& name, hipFunction_t* fn) {
    std::lock_guard<std::mutex> lk(g_gemm4_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        scail_set_error("gemm4: hipGetDevice failed");
        return 2;
    }
    const bool is8 = name.rfind("scail_gemm8", 0) == 0;
    auto mit = g_gemm4_modules.find(std::make_pair(dev, is8 ? 8 : 4));
    if (mit == g_gemm4_modules.end()) {
#ifdef SCAIL_ABLATIONS
        const void* image = is8 ? (const void*)k_gemm8_hsaco : (const void*)k_gemm4_hsaco;
#else
        const void* image = k_gemm4_hsaco;
#endif
        hipModule_t mod = nullptr;
        hipError_t e = hipModuleLoadData(&mod, image);
        if (e != hipSuccess) {
            scail_set_error(std::string("gemm4: hipModuleLoadData failed: ") + hipGetErrorString(e));
            return 2;
        }
        mit = g_gemm4_modules.emplace(std::make_pair(dev, is8 ? 8 : 4), mod).first;
    }
    auto it = g_gemm4_fn.find(std::make_pair(dev, name));
    if (it == g_gemm4_fn.end()) {
        hipFunction_t f;
        hipError_t e = hipModuleGetFunction(&f, mit->second, name.c_str());
        if (e != hipSuccess) {
            scail_set_error("gemm4: kernel " + name + " is not in the embedded code object: " + hipGetErrorString(e));
            return 2;
        }
        it = g_gemm4_fn.emplace(std::make_pair(dev, name), f).first;
    }
    *fn = it->second;
    return 0;
}

// Tile order table: workgroup id b runs on XCD b % 8 and takes entry b >> 3 of that XCD's sequence; a sequence is made of whole tile
// groups (group_m m-tiles x one n-tile at a time), so the 32 tiles in flight on an XCD share ~4 + 8 operand panels in its L2.
static int gemm4_table(int tm, int tn, int group_m, uint32_t** dev_table, int* entries) {
    std::lock_guard<std::mutex> lk(g_gemm4_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        scail_set_error("gemm4: hipGetDevice failed");
        return 2;
    }
    const int table_mode = g_gemm4_table_mode;      // read once, under the mutex (scail_tune_set writes it under the same mutex)
    auto key = std::make_tuple(dev, tm, tn, group_m, table_mode);
    auto it = g_gemm4_tables.find(key);
    if (it == g_gemm4_tables.end()) {
        // per-XCD tile sequences (workgroup b runs on XCD b % 8 and takes entry b >> 3 of that XCD's sequence)
        std::vector<std::vector<uint32_t>> seq(8);
        auto tile = [](int m, int n) { return (uint32_t)m | ((uint32_t)n << 16); };
        const int n_groups = (tm + group_m - 1) / group_m;
        if (table_mode == 0) {
            // (round 1 / 2, measurement build) every XCD walks a contiguous range of the grouped order: 8 distant m-regions
            std::vector<uint32_t> order;
            for (int g0 = 0; g0 < tm; g0 += group_m)
                for (int n = 0; n < tn; ++n)
                    for (int m = g0; m < std::min(g0 + group_m, tm); ++m) order.push_back(tile(m, n));
            const int T = (int)order.size(), per = (T + 7) / 8;
            for (int x = 0; x < 8; ++x)
                for (int i = x * per; i < std::min((x + 1) * per, T); ++i) seq[x].push_back(order[i]);
        } else if (table_mode == 1) {
            // default (round 3): m-groups dealt round-robin: the 8 XCDs work on 8 ADJACENT m-groups and sweep n together, so a W
            // panel is wanted by all XCDs at about the same time (one HBM fetch, seven Infinity-Cache hits): +1.0-1.8 % on the qkv /
            // MLP shapes over the contiguous ranges (profiles/r03_gemm_table_modes.log), neutral on the N = 5120 ones.
            // Only WHOLE rows of 8 groups are dealt; the tiles of the remaining < 8 groups are split evenly over the XCDs in grouped
            // order (dealing 6 or 12 groups -- the 6 104- / 12 208-row GEMMs of one sequence-parallel rank at 8 ranks -- would leave
            // XCDs idle or doubly loaded: measured 262 instead of ~220 ms of GEMM time per step, profiles/r03_sp8_kernel_stats.md)
            const int dealt = n_groups / 8 * 8;
            for (int g = 0; g < dealt; ++g)
                for (int n = 0; n < tn; ++n)
                    for (int m = g * group_m; m < std::min((g + 1) * group_m, tm); ++m) seq[g % 8].push_back(tile(m, n));
            std::vector<uint32_t> rest;
            for (int g = dealt; g < n_groups; ++g)
                for (int n = 0; n < tn; ++n)
                    for (int m = g * group_m; m < std::min((g + 1) * group_m, tm); ++m) rest.push_back(tile(m, n));
            const int T = (int)rest.size(), per = (T + 7) / 8;
            for (int x = 0; x < 8; ++x)
                for (int i = x * per; i < std::min((x + 1) * per, T); ++i) seq[x].push_back(rest[i]);
        } else {
            // (measurement build) XCD pairs share an m-group and split its n sweep in halves: 16 instead of 32 live x panels
            const int half = (tn + 1) / 2;
            for (int g = 0; g < n_groups; ++g)
                for (int hx = 0; hx < 2; ++hx)
                    for (int n = hx * half; n < std::min((hx + 1) * half, tn); ++n)
                        for (int m = g * group_m; m < std::min((g + 1) * group_m, tm); ++m) seq[2 * (g % 4) + hx].push_back(tile(m, n));
        }
        size_t per = 0;
        for (auto& q : seq) per = std::max(per, q.size());
        std::vector<uint32_t> table(per * 8);
        for (size_t b = 0; b < per * 8; ++b) table[b] = (b >> 3) < seq[b & 7].size() ? seq[b & 7][b >> 3] : 0xFFFFFFFFu;
        uint32_t* d = nullptr;
        if (hipMalloc(&d, table.size() * 4) != hipSuccess || hipMemcpy(d, table.data(), table.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            scail_set_error("gemm4: cannot allocate the tile order table (the first call for a tile grid allocates; do it outside stream capture)");
            return 2;
        }
        it = g_gemm4_tables.emplace(key, std::make_pair(d, (int)(per * 8))).first;
    }
    *dev_table = it->second.first;
    *entries = it->second.second;
    return 0;
}

static int gemm4_cu_count() {
    static std::map<int, int> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_gemm4_mutex);
    auto it = cache.find(dev);
    if (it == cache.end()) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        it = cache.emplace(dev, n).first;
    }
    return it->second;
}

// Frees the tile-order tables of EVERY device (scail_release_caches, include/scail_hip.h): call with no launch in flight.  A table's
// device pointer is a kernel argument of every gemm4 launch, so it is baked into any hipGraph captured from such launches: graphs
// captured before this call must be destroyed or re-captured, never replayed.
int scail_gemm4_release_tables() {
    std::lock_guard<std::mutex> lk(g_gemm4_mutex);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& kv : g_gemm4_tables) {
        (void)hipSetDevice(std::get<0>(kv.first));
        (void)hipFree(kv.second.first);
    }
    g_gemm4_tables.clear();
    (void)hipSetDevice(cur);
    return 0;
}

static bool gemm4_eligible(int64_t lda, int64_t ldc, int64_t ldr, int64_t M, int64_t N, int64_t K, int epilogue) {
    const int64_t lim = 1ll << 31;
    const int64_t tail = M % 256;
    // M >= 2048: always; 512 <= M < 2048 (round 5: the text K / V projections of the conditioning, 1024 x 10240 x 5120): where the 256 x 256 tiling
    // still yields at least half a round of tiles (the 128 x 128 hipcc kernel keeps the shapes with few tiles)
    const bool rows = M >= 2048 || (M >= 512 && (M + 255) / 256 * (N / 256) >= 128);
    return rows && (tail == 0 || tail >= 8) && N % 256 == 0 && K % 64 == 0 && K >= 128 &&
           (epilogue == SCAIL_EPI_BIAS || epilogue == SCAIL_EPI_GELU_TANH || epilogue == SCAIL_EPI_RESID) &&
           M * ldc < lim && M * std::max<int64_t>(ldr, 1) < lim && 256 * lda < lim;
}

extern "C" int scail_gemm_kernel_for(int64_t lda, int64_t ldc, int64_t ldr, int64_t M, int64_t N, int64_t K, int epilogue) {
    const int mode = g_gemm4_mode;
    return (mode && gemm4_eligible(lda, ldc, ldr, M, N, K, epilogue)) ? mode : 0;
}

// load the embedded code object and resolve the four shipped kernels now (see scail_attn4_preload)
int scail_gemm4_preload() {
    hipFunction_t fn;
    for (int e : {0, 1, 3, 4})
        if (int rc = gemm4_function("scail_gemm4_e" + std::to_string(e), &fn)) return rc;
    return 0;
}

// runtime option of the product library (scail_set_option "gemm4"): 1 = generated kernels where eligible, 0 = csrc/gemm.hip only
int scail_gemm4_enable(int on) { g_gemm4_mode = on ? 4 : 0; return 0; }

#ifdef SCAIL_ABLATIONS
int scail_gemm4_knob(const char* knob, int value) {
    std::string k(knob);
    std::lock_guard<std::mutex> lk(g_gemm4_mutex);      // gemm4_table / the launch read these under the same mutex
    if (k == "gemm4") { g_gemm4_mode = (value == 4 || value == 8) ? value : (value ? 4 : 0); return 0; }
    if (k == "gemm4_table") { g_gemm4_table_mode = (value >= 0 && value <= 2) ? value : 1; return 0; }
    if (k.rfind("gemm4_kernel", 0) == 0) { g_gemm4_suffix = k.size() > 13 ? "_" + k.substr(13) : ""; return 0; }
    return -1;
}
#endif

extern "C" int scail_gemm_bf16(const scail_bf16* x, int64_t lda, const scail_bf16* w, const float* bias,
                               scail_bf16* y, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                               const scail_bf16* resid, int64_t ldr, const float* gate, int64_t gate_stride,
                               int64_t rows_per_batch, void* stream) {
    SCAIL_REQUIRE(K > 0 && K % BK == 0, "K must be a positive multiple of 64");
    SCAIL_REQUIRE(N % 8 == 0, "N must be a multiple of 8");
    SCAIL_REQUIRE(lda % 8 == 0 && ldc % 4 == 0, "lda must be a multiple of 8, ldc of 4");
    SCAIL_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "dimension too large");
    SCAIL_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(y) & 7) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0,
                  "pointer alignment");
    if (M == 0 || N == 0) return 0;
    if (epilogue == SCAIL_EPI_RESID) {
        SCAIL_REQUIRE(resid != nullptr && ldr % 4 == 0, "RESID epilogue needs resid with ldr % 4 == 0");
        SCAIL_REQUIRE(gate == nullptr || (rows_per_batch > 0 && gate_stride % 4 == 0), "gate needs rows_per_batch > 0, gate_stride % 4 == 0");
    }
#ifdef SCAIL_ABLATIONS
    const bool forced_tile = g_gemm_tile != 0;
#else
    const bool forced_tile = false;
#endif
    if (g_gemm4_mode && !forced_tile && gemm4_eligible(lda, ldc, ldr, M, N, K, epilogue)) {
        const int epi4 = epilogue == SCAIL_EPI_RESID ? (gate != nullptr ? 3 : 4) : epilogue;
        const bool is8 = g_gemm4_mode == 8;
        std::string name = std::string(is8 ? "scail_gemm8_e" : "scail_gemm4_e") + std::to_string(epi4);
        if (epi4 == 0 || g_gemm4_suffix == "_pst" || g_gemm4_suffix == "_part" || g_gemm4_suffix == "_stgnt") name += g_gemm4_suffix;      // A/B variants: bias epilogue only, except the persistent set (ablation build)
        hipFunction_t fn;
        if (int rc = gemm4_function(name, &fn)) return rc;
        uint32_t* table;
        int entries;
        if (int rc = gemm4_table((int)((M + 255) / 256), (int)(N / 256), g_group_m, &table, &entries)) return rc;
        Gemm4Args a;
        a.x = x; a.w = w; a.bias = bias; a.y = y; a.resid = resid; a.gate = gate; a.table = table;
        a.lda = lda; a.ldc = ldc; a.ldr = ldr; a.gs = gate_stride;
        a.M = (int32_t)M; a.N = (int32_t)N; a.K = (int32_t)K; a.rows_per_batch = (int32_t)rows_per_batch;
        // persistent kernels (name ends in "_pst"): one workgroup per CU, rounded down to a multiple of 8 so that workgroup b keeps
        // walking the sequence of XCD b % 8 (entries b, b + grid, b + 2 grid, ...)
        unsigned grid = (unsigned)entries;
        const bool persistent = name.size() > 4 && name.compare(name.size() - 4, 4, "_pst") == 0;
        if (persistent) {
            int cus = gemm4_cu_count();
            if (cus < 8) cus = 8;
            grid = std::min<unsigned>((unsigned)entries, (unsigned)(cus / 8 * 8));
        }
        a.grid = (int32_t)grid; a.entries = (int32_t)entries;
        size_t sz = sizeof(a);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        hipError_t e = hipModuleLaunchKernel(fn, grid, 1, 1, is8 ? 512 : 256, 1, 1, 0, (hipStream_t)stream, nullptr, extra);
        if (e != hipSuccess) {
            scail_set_error(std::string("gemm4: launch failed: ") + hipGetErrorString(e));
            return 2;
        }
        return 0;
    }
    GemmParams p;
    p.x = x; p.lda = lda; p.w = w; p.bias = bias; p.y = y; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.resid = resid; p.ldr = ldr; p.gate = gate; p.gate_stride = gate_stride; p.rows_per_batch = rows_per_batch;
    p.group_m = g_group_m;
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case SCAIL_EPI_BIAS: return launch_gemm<SCAIL_EPI_BIAS>(p, s);
        case SCAIL_EPI_GELU_TANH: return launch_gemm<SCAIL_EPI_GELU_TANH>(p, s);
        case SCAIL_EPI_GELU_ERF: return launch_gemm<SCAIL_EPI_GELU_ERF>(p, s);
        case SCAIL_EPI_RESID:
            SCAIL_REQUIRE(resid != nullptr && ldr % 4 == 0, "RESID epilogue needs resid with ldr % 4 == 0");
            SCAIL_REQUIRE(gate == nullptr || (rows_per_batch > 0 && gate_stride % 4 == 0), "gate needs rows_per_batch > 0, gate_stride % 4 == 0");
            return launch_gemm<SCAIL_EPI_RESID>(p, s);
        default:
            scail_set_error("scail_gemm_bf16: unknown epilogue");
            return 1;
    }
}

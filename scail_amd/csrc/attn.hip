// Flash attention (no mask, head_dim 128) for the SCAIL DiT on gfx950: self-attention over the
// concatenated [ref | noise | pose] token sequence (L = 48 832 at 512p x 81 f) and the two short
// cross-attentions (512 text / 257 CLIP keys).
//
// Work decomposition: one 512-thread workgroup (8 waves) per (256 query rows, head, batch); each
// wave owns 32 query rows.  K and V^T tiles of 64 keys are staged through registers into a
// double-buffered LDS ring shared by the 8 waves (one barrier per tile; the global loads of tile
// t+1 are issued before the MFMAs of tile t and written to the other buffer after them).
//
// MFMA formulation (v_mfma_f32_32x32x16_bf16), chosen so that NO cross-lane data movement is
// needed between the two GEMMs:
//   S^T = K . Q^T    A = K fragment  (rows = keys, from LDS), B = Q fragment (registers, loaded once)
//                    -> lane (q = lane & 31, g = lane >> 5) holds, for its query row q, the scores of
//                       keys 32 f + (r & 3) + 8 (r >> 2) + 4 g,  f = 0..1, r = 0..15
//   O^T += V^T . P^T A = V^T fragment (rows = d, from LDS),  B = P^T fragment = the SAME lane's 8
//                       consecutive score registers packed to bf16.  The MFMA contraction only needs
//                       A and B to agree on which key sits in k-slot (g, j); V^T is stored with key
//                       bits 2 and 3 swapped inside each 16-key group (scail_transpose_v), which makes
//                       the 8 keys of slot group (ks, g) one contiguous 16-byte LDS read.
// Online softmax in fp32 (running max m, running sum l per query row, exp2 with the scale folded
// in); a row's two lanes (g = 0, 1) keep the same m and separate partial l that are added once at
// the end.  LDS rows are padded (K: 272 B, V^T: 144 B) so every ds_read_b128 service group touches
// 16 distinct 16-byte slots.
#include <atomic>
#include <map>
#include <mutex>
#include "common.h"

#define HD 128
#define QBLK 256
#define KVBLK 64
#define ATT_THREADS 512
#define K_LD (HD + 8)      // elements per K row in LDS (272 B)
#define V_LD (KVBLK + 8)   // elements per V^T row in LDS (144 B)
#define ATT_LDS_BYTES (2 * (KVBLK * K_LD + HD * V_LD) * 2)
#define ATT_PP_LDS_BYTES (ATT_LDS_BYTES + QBLK * K_LD * 2)

struct AttnParams {
    const u16* q; int64_t q_bs, q_rs;
    const u16* k; int64_t k_ss, k_bs, k_rs;
    const u16* vt; int64_t vt_ss, vt_bs;
    u16* o; int64_t o_bs, o_rs;
    int heads, Lq, Lk, Lkp, n_seg;
    float sl2;  // scale * log2(e)
    int accumulate;
    int probe;  // measurement aid: add this workgroup's lifetime (s_memtime ticks) to g_attn_clk
};

__device__ unsigned long long g_attn_clk[2] = {0ull, 0ull};
#ifdef SCAIL_ABLATIONS
int scail_attn_clk(unsigned long long* out2, int reset) {
    if (out2 != nullptr) {
        unsigned long long v[2];
        if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_attn_clk), 16) != hipSuccess) return 2;
        out2[0] += v[0]; out2[1] += v[1];
    }
    if (reset) {
        const unsigned long long z[2] = {0ull, 0ull};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_clk), z, 16) != hipSuccess) return 2;
    }
    return 0;
}
#endif

// VARIANT bit 0: s_setprio(1) around the MFMA clusters; bit 1: skip the O rescale when no row's running
// max moved in this tile (exact: alpha == 1 for every lane).
// ABLATION ONLY (wrong results, tools/microbench.py): bit 4 = no softmax (P = bf16(S)), bit 5 = no K/V
// staging and no barrier inside the loop (tile 0 reused).
// NW = waves per workgroup: 8 (256 query rows, one workgroup per CU) or 4 (128 rows, TWO independent
// workgroups per CU: while one sits in its per-tile barrier / first-LDS-read bubble or in its softmax,
// the other keeps the SIMDs' matrix pipes busy; costs 2x the L2->LDS K/V traffic).
template <int VARIANT, int NW>
__global__ __launch_bounds__(64 * NW, 2) void flash_attn_kernel(AttnParams p) {
    constexpr int NTH = 64 * NW;
    // bit 8: K / V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, written
    // lane-linearly, so rows cannot be padded); bank conflicts are removed by an XOR swizzle of the 16-byte
    // chunk index applied to the per-lane SOURCE address and to the fragment reads (same involution).
    constexpr bool DMA = (VARIANT & 256) != 0;
    constexpr int KLD = DMA ? HD : K_LD, VLD = DMA ? KVBLK : V_LD;
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* Ks = smem;                     // [2][KVBLK][KLD]
    u16* Vs = smem + 2 * KVBLK * KLD;   // [2][HD][VLD]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, g = lane >> 5;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int q0 = blockIdx.x * (32 * NW) + wave * 32;
    const unsigned long long clk0 = p.probe ? __builtin_amdgcn_s_memtime() : 0ull;

    // ---- Q fragments (B operand of S^T = K Q^T): Q[q][16 ks + 8 g .. +7] ----
    bf16x8 qf[HD / 16];
    {
        const int qrow = min(q0 + ql, p.Lq - 1);
        const u16* qp = p.q + b * p.q_bs + (int64_t)qrow * p.q_rs + (int64_t)h * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    // ---- staging coordinates: 1024 K chunks + 1024 V^T chunks of 16 B per tile, 1024 / NTH each per thread ----
    constexpr int KRS = NTH / 16, VRS = NTH / 8;   // tile rows covered per staging pass
    const int krow = tid >> 4, kcc = tid & 15;     // K chunk c = tid + NTH i: row = (tid >> 4) + KRS i
    const int vrow = tid >> 3, vcc = tid & 7;      // V chunk c = tid + NTH i: row = (tid >> 3) + VRS i
    const u16* kbase = p.k + b * p.k_bs + (int64_t)h * HD + kcc * 8;
    const u16* vbase = p.vt + b * p.vt_bs + (int64_t)h * HD * p.Lkp + vcc * 8;
    const int tps = p.Lkp / KVBLK;          // tiles per segment
    const int ntiles = tps * p.n_seg;
    uint4 kr0, kr1, kr2, kr3, vr0, vr1, vr2, vr3;  // named scalars (register arrays written under a branch go to scratch)
#define G_LOAD(tt_)                                                                               \
    {                                                                                             \
        const int seg_ = (tt_) / tps, t_ = (tt_) - seg_ * tps;                                    \
        const int key0_ = t_ * KVBLK;                                                             \
        const u16* kp_ = kbase + (int64_t)seg_ * p.k_ss;                                          \
        const u16* vp_ = vbase + (int64_t)seg_ * p.vt_ss + key0_;                                 \
        kr0 = *reinterpret_cast<const uint4*>(kp_ + (int64_t)min(key0_ + krow, p.Lk - 1) * p.k_rs);            \
        kr1 = *reinterpret_cast<const uint4*>(kp_ + (int64_t)min(key0_ + krow + KRS, p.Lk - 1) * p.k_rs);      \
        vr0 = *reinterpret_cast<const uint4*>(vp_ + (int64_t)vrow * p.Lkp);                       \
        vr1 = *reinterpret_cast<const uint4*>(vp_ + (int64_t)(vrow + VRS) * p.Lkp);               \
        if (NW == 4) {                                                                            \
            kr2 = *reinterpret_cast<const uint4*>(kp_ + (int64_t)min(key0_ + krow + 2 * KRS, p.Lk - 1) * p.k_rs);  \
            kr3 = *reinterpret_cast<const uint4*>(kp_ + (int64_t)min(key0_ + krow + 3 * KRS, p.Lk - 1) * p.k_rs);  \
            vr2 = *reinterpret_cast<const uint4*>(vp_ + (int64_t)(vrow + 2 * VRS) * p.Lkp);       \
            vr3 = *reinterpret_cast<const uint4*>(vp_ + (int64_t)(vrow + 3 * VRS) * p.Lkp);       \
        }                                                                                         \
    }
#define S_STORE(buf_)                                                                             \
    {                                                                                             \
        *reinterpret_cast<uint4*>(Ks + ((buf_) * KVBLK + krow) * K_LD + kcc * 8) = kr0;           \
        *reinterpret_cast<uint4*>(Ks + ((buf_) * KVBLK + krow + KRS) * K_LD + kcc * 8) = kr1;     \
        *reinterpret_cast<uint4*>(Vs + ((buf_) * HD + vrow) * V_LD + vcc * 8) = vr0;              \
        *reinterpret_cast<uint4*>(Vs + ((buf_) * HD + vrow + VRS) * V_LD + vcc * 8) = vr1;        \
        if (NW == 4) {                                                                            \
            *reinterpret_cast<uint4*>(Ks + ((buf_) * KVBLK + krow + 2 * KRS) * K_LD + kcc * 8) = kr2;  \
            *reinterpret_cast<uint4*>(Ks + ((buf_) * KVBLK + krow + 3 * KRS) * K_LD + kcc * 8) = kr3;  \
            *reinterpret_cast<uint4*>(Vs + ((buf_) * HD + vrow + 2 * VRS) * V_LD + vcc * 8) = vr2;     \
            *reinterpret_cast<uint4*>(Vs + ((buf_) * HD + vrow + 3 * VRS) * V_LD + vcc * 8) = vr3;     \
        }                                                                                         \
    }

    // LDS-DMA pieces of this wave: K piece j = rows 4j..4j+3 (lane -> row 4j + (l >> 4), chunk l & 15),
    // V^T piece j = rows 8j..8j+7 (lane -> row 8j + (l >> 3), chunk l & 7); 16 pieces each per tile.
    constexpr int PPW = 16 / NW;   // pieces per wave and operand
    const int dk_row = lane >> 4, dk_c = lane & 15, dv_row = lane >> 3, dv_c = lane & 7;
#define DMA_ISSUE(tt_, buf_)                                                                                \
    {                                                                                                       \
        const int seg_ = (tt_) / tps, t_ = (tt_) - seg_ * tps;                                              \
        const int key0_ = t_ * KVBLK;                                                                       \
        const u16* kp_ = p.k + b * p.k_bs + (int64_t)h * HD + (int64_t)seg_ * p.k_ss;                       \
        const u16* vp_ = p.vt + b * p.vt_bs + (int64_t)h * HD * p.Lkp + (int64_t)seg_ * p.vt_ss + key0_;    \
        _Pragma("unroll") for (int i_ = 0; i_ < PPW; ++i_) {                                                \
            const int j_ = wave * PPW + i_;                                                                 \
            const int kr_ = 4 * j_ + dk_row;                                                                \
            const u16* ks_src = kp_ + (int64_t)min(key0_ + kr_, p.Lk - 1) * p.k_rs + ((dk_c ^ (kr_ & 15)) << 3);  \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ks_src,          \
                (__attribute__((address_space(3))) void*)(Ks + ((buf_) * KVBLK + 4 * j_) * KLD), 16, 0, 0);  \
            const int vr_ = 8 * j_ + dv_row;                                                                \
            const u16* vs_src = vp_ + (int64_t)vr_ * p.Lkp + ((dv_c ^ ((vr_ >> 1) & 7)) << 3);              \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vs_src,          \
                (__attribute__((address_space(3))) void*)(Vs + ((buf_) * HD + 8 * j_) * VLD), 16, 0, 0);     \
        }                                                                                                   \
    }
    // swizzled fragment-read offsets (elements): chunk (2 ks + g) of row ql
    int koff[HD / 16], voff[4];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) koff[ks] = DMA ? (((2 * ks + g) ^ (ql & 15)) << 3) : (ks * 16 + g * 8);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) voff[ks] = DMA ? (((2 * ks + g) ^ ((ql >> 1) & 7)) << 3) : (ks * 16 + g * 8);

    f32x16 o[HD / 32];
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2 = p.sl2;
    const int tail = p.Lk - (tps - 1) * KVBLK;  // valid keys in the last tile of a segment (1..64)

    if (DMA) {
        DMA_ISSUE(0, 0)
    } else {
        G_LOAD(0)
    }
    // drain the Q-fragment loads here: otherwise hipcc's in-order vmcnt bookkeeping makes the first
    // QK^T MFMAs of EVERY iteration wait for the freshly issued tile t+1 loads (vmcnt(3..0))
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
    if (!DMA) S_STORE(0)
    __syncthreads();
    for (int tt = 0; tt < ntiles; ++tt) {
        const int cur = (VARIANT & 32) ? 0 : (tt & 1);
        if (!(VARIANT & 32)) {
            if (tt + 1 < ntiles) {
                if (DMA) {
                    DMA_ISSUE(tt + 1, cur ^ 1)
                } else {
                    G_LOAD(tt + 1)
                }
            }
        }

        // ---- S^T = K Q^T ----
        f32x16 s[2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[f][e] = 0.f;
        const u16* ks_ = Ks + (cur * KVBLK + ql) * KLD;
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks_ + f * 32 * KLD + koff[ks]);
                s[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[f], 0, 0, 0);
            }
        }
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
        // ---- mask the padded keys of a segment's last tile ----
        if (tail < KVBLK) {
            const int t = tt % tps;
            if (t == tps - 1) {
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * f + (r & 3) + 8 * (r >> 2) + 4 * g;
                        if (key >= tail) s[f][r] = -INFINITY;
                    }
            }
        }
        // ---- online softmax ----
        if (!(VARIANT & 16)) {
        float mx = s[0][0];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[f][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
        const float msc = m_new * sl2;
        float rs = 0.f;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[f][r] * sl2 - msc);
                s[f][r] = pv;
                rs += pv;
            }
        l_run = l_run * alpha + rs;
        if (!(VARIANT & 2) || !__all(m_new == m_run)) {
#pragma unroll
            for (int d = 0; d < HD / 32; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
        }
        m_run = m_new;
        } else { l_run = 1.f; }
        // ---- P^T fragments: k-slot group ks <-> score registers s[ks>>1][8 (ks&1) .. +7] ----
        bf16x8 pf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 u;
            const int f = ks >> 1, r0 = (ks & 1) * 8;
            u.x = pack_bf16x2(s[f][r0 + 0], s[f][r0 + 1]);
            u.y = pack_bf16x2(s[f][r0 + 2], s[f][r0 + 3]);
            u.z = pack_bf16x2(s[f][r0 + 4], s[f][r0 + 5]);
            u.w = pack_bf16x2(s[f][r0 + 6], s[f][r0 + 7]);
            pf[ks] = __builtin_bit_cast(bf16x8, u);
        }
        // ---- O^T += V^T P^T ----
        const u16* vs_ = Vs + (cur * HD + ql) * VLD;
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int d = 0; d < HD / 32; ++d) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vs_ + d * 32 * VLD + voff[ks]);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], o[d], 0, 0, 0);
            }
        }
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
        if (!(VARIANT & 32)) {
            if (!DMA) { if (tt + 1 < ntiles) { S_STORE(cur ^ 1) } }
            __syncthreads();
        }
    }

    // ---- epilogue: O[q][32 d + 8 rr + 4 g + e] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + ql;
    if (qrow < p.Lq) {
        u16* op = p.o + b * p.o_bs + (int64_t)qrow * p.o_rs + (int64_t)h * HD + 4 * g;
#pragma unroll
        for (int d = 0; d < HD / 32; ++d) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[d][4 * rr + e] * inv;
                uint2* dst = reinterpret_cast<uint2*>(op + d * 32 + rr * 8);
                if (p.accumulate) {
                    const uint2 old = *dst;
                    v[0] += bf_lo(old.x); v[1] += bf_hi(old.x);
                    v[2] += bf_lo(old.y); v[3] += bf_hi(old.y);
                }
                uint2 w;
                w.x = pack_bf16x2(v[0], v[1]);
                w.y = pack_bf16x2(v[2], v[3]);
                *dst = w;
            }
        }
    }
    if (p.probe && tid == 0) {
        atomicAdd(&g_attn_clk[0], (unsigned long long)__builtin_amdgcn_s_memtime() - clk0);
        atomicAdd(&g_attn_clk[1], 1ull);
    }
}


// ================================================================================================
// Cross attention over TWO short key sets in one launch: O = softmax(q K1^T) V1 + softmax(q K2^T) V2 (text + CLIP image
// tokens, dit_video_crossattn_sc_xc.py:1107-1203: two scaled-dot-product attentions over the same queries whose bf16
// outputs are added).  The two sets keep independent softmax states: when the tile sequence crosses from set 1 to set 2 the
// normalised O1 is packed to bf16 (32 registers; the reference rounds each attention output to bf16 before the add as well),
// the state is reset, and the epilogue writes bf16(O1) + O2 -- Q is read once, O written once, nothing is read back.
// Launch shape for SHORT key sets (13 tiles at 512 + 257 keys): 4 waves x 32 query rows, 64 KB of LDS -> two workgroups per
// CU, so one workgroup's prologue (Q + first tile: ~2 us of latency against ~20 us of work), softmax and barriers overlap
// the other's MFMAs; the K / V^T tiles of a (batch, head) are shared by its 382 workgroups and stay in L2.
// ================================================================================================
struct Cross2Params {
    const u16* q; int64_t q_bs, q_rs;
    const u16* k0; int64_t k0_bs, k0_rs; const u16* vt0; int64_t vt0_bs; int Lk0, Lkp0;
    const u16* k1; int64_t k1_bs, k1_rs; const u16* vt1; int64_t vt1_bs; int Lk1, Lkp1;
    u16* o; int64_t o_bs, o_rs;
    int heads, Lq;
    float sl2;  // scale * log2(e)
};

#define X2_THREADS 256
#define X2_LDS_BYTES (2 * (KVBLK * HD + HD * KVBLK) * 2)   // 64 KiB: two workgroups per CU
__global__ __launch_bounds__(X2_THREADS, 2) void cross_attn2_kernel(Cross2Params p) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* Ks = smem;                      // [2][KVBLK][HD]
    u16* Vs = smem + 2 * KVBLK * HD;     // [2][HD][KVBLK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, g = lane >> 5;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;

    // ---- Q fragments (B operand of S^T = K Q^T): Q[q][16 ks + 8 g .. +7] ----
    bf16x8 qf[HD / 16];
    {
        const int qrow = min(q0 + ql, p.Lq - 1);
        const u16* qp = p.q + b * p.q_bs + (int64_t)qrow * p.q_rs + (int64_t)h * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    // ---- staging by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, lane-linear in LDS, no registers): per tile
    // 16 K pieces of 4 rows + 16 V^T pieces of 8 rows, 4 + 4 per wave.  Rows are unpadded; the 16-byte chunk index is XOR-swizzled
    // on the per-lane SOURCE address and in the fragment reads (same involution), which keeps the reads conflict-free.
    const int dk_row = lane >> 4, dk_c = lane & 15, dv_row = lane >> 3, dv_c = lane & 7;
    const u16* kb0 = p.k0 + b * p.k0_bs + (int64_t)h * HD;
    const u16* kb1 = p.k1 + b * p.k1_bs + (int64_t)h * HD;
    const u16* vb0 = p.vt0 + b * p.vt0_bs + (int64_t)h * HD * p.Lkp0;
    const u16* vb1 = p.vt1 + b * p.vt1_bs + (int64_t)h * HD * p.Lkp1;
    const int n0 = p.Lkp0 / KVBLK, n1 = p.Lkp1 / KVBLK, ntiles = n0 + n1;
#define X2_ISSUE(tt_, buf_)                                                                                 \
    {                                                                                                       \
        const bool s1_ = (tt_) >= n0;                                                                       \
        const int key0_ = ((tt_) - (s1_ ? n0 : 0)) * KVBLK;                                                 \
        const u16* kp_ = s1_ ? kb1 : kb0;                                                                   \
        const int64_t krs_ = s1_ ? p.k1_rs : p.k0_rs;                                                       \
        const int last_ = (s1_ ? p.Lk1 : p.Lk0) - 1;                                                        \
        const int lkp_ = s1_ ? p.Lkp1 : p.Lkp0;                                                             \
        const u16* vp_ = (s1_ ? vb1 : vb0) + key0_;                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                  \
            const int j_ = wave * 4 + i_;                                                                   \
            const int kr_ = 4 * j_ + dk_row;                                                                \
            const u16* ks_src = kp_ + (int64_t)min(key0_ + kr_, last_) * krs_ + ((dk_c ^ (kr_ & 15)) << 3); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ks_src,          \
                (__attribute__((address_space(3))) void*)(Ks + ((buf_) * KVBLK + 4 * j_) * HD), 16, 0, 0);   \
            const int vr_ = 8 * j_ + dv_row;                                                                \
            const u16* vs_src = vp_ + (int64_t)vr_ * lkp_ + ((dv_c ^ ((vr_ >> 1) & 7)) << 3);               \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vs_src,          \
                (__attribute__((address_space(3))) void*)(Vs + ((buf_) * HD + 8 * j_) * KVBLK), 16, 0, 0);   \
        }                                                                                                   \
    }
    // swizzled fragment-read offsets (elements): chunk (2 ks + g) of row ql
    int koff[HD / 16], voff[4];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) koff[ks] = ((2 * ks + g) ^ (ql & 15)) << 3;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) voff[ks] = ((2 * ks + g) ^ ((ql >> 1) & 7)) << 3;

    f32x16 o[HD / 32];
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    uint2 o1[HD / 32][4];                     // bf16(O1) of key set 1, in the epilogue's store layout
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2 = p.sl2;

    X2_ISSUE(0, 0)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): tile 0 has landed in LDS, Q fragments in registers
    __syncthreads();
    for (int tt = 0; tt < ntiles; ++tt) {
        const int cur = tt & 1;
        if (tt + 1 < ntiles) X2_ISSUE(tt + 1, cur ^ 1)
        if (tt == n0) {
            // ---- key set 1 is complete: O1 = acc / l, rounded to bf16; fresh softmax state for set 2 ----
            const float inv1 = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
#pragma unroll
            for (int d = 0; d < HD / 32; ++d)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    o1[d][rr].x = pack_bf16x2(o[d][4 * rr + 0] * inv1, o[d][4 * rr + 1] * inv1);
                    o1[d][rr].y = pack_bf16x2(o[d][4 * rr + 2] * inv1, o[d][4 * rr + 3] * inv1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[d][4 * rr + e] = 0.f;
                }
            m_run = -INFINITY;
            l_run = 0.f;
        }
        // ---- S^T = K Q^T ----
        f32x16 s[2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[f][e] = 0.f;
        const u16* ks_ = Ks + (cur * KVBLK + ql) * HD;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks_ + f * 32 * HD + koff[ks]);
                s[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[f], 0, 0, 0);
            }
        }
        // ---- mask the padded keys of each set's last tile ----
        {
            const bool s1 = tt >= n0;
            const int t = tt - (s1 ? n0 : 0), ns = s1 ? n1 : n0;
            const int tail = (s1 ? p.Lk1 : p.Lk0) - (ns - 1) * KVBLK;   // valid keys in the set's last tile (1..64)
            if (t == ns - 1 && tail < KVBLK) {
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * f + (r & 3) + 8 * (r >> 2) + 4 * g;
                        if (key >= tail) s[f][r] = -INFINITY;
                    }
            }
        }
        // ---- online softmax ----
        float mx = s[0][0];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[f][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
        const float msc = m_new * sl2;
        float rs = 0.f;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[f][r] * sl2 - msc);
                s[f][r] = pv;
                rs += pv;
            }
        l_run = l_run * alpha + rs;
        if (!__all(m_new == m_run)) {        // exact skip: alpha == 1 for every lane
#pragma unroll
            for (int d = 0; d < HD / 32; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
        }
        m_run = m_new;
        // ---- P^T fragments: k-slot group ks <-> score registers s[ks>>1][8 (ks&1) .. +7] ----
        bf16x8 pf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 u;
            const int f = ks >> 1, r0 = (ks & 1) * 8;
            u.x = pack_bf16x2(s[f][r0 + 0], s[f][r0 + 1]);
            u.y = pack_bf16x2(s[f][r0 + 2], s[f][r0 + 3]);
            u.z = pack_bf16x2(s[f][r0 + 4], s[f][r0 + 5]);
            u.w = pack_bf16x2(s[f][r0 + 6], s[f][r0 + 7]);
            pf[ks] = __builtin_bit_cast(bf16x8, u);
        }
        // ---- O^T += V^T P^T ----
        const u16* vs_ = Vs + (cur * HD + ql) * KVBLK;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int d = 0; d < HD / 32; ++d) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vs_ + d * 32 * KVBLK + voff[ks]);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], o[d], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of tile tt + 1 are in LDS before the barrier
        __syncthreads();
    }

    // ---- epilogue: O[q][32 d + 8 rr + 4 g + e] = bf16(O1) + O2 ----
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    const int qrow = q0 + ql;
    if (qrow < p.Lq) {
        u16* op = p.o + b * p.o_bs + (int64_t)qrow * p.o_rs + (int64_t)h * HD + 4 * g;
#pragma unroll
        for (int d = 0; d < HD / 32; ++d) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                uint2 w;
                w.x = pack_bf16x2(o[d][4 * rr + 0] * inv + bf_lo(o1[d][rr].x), o[d][4 * rr + 1] * inv + bf_hi(o1[d][rr].x));
                w.y = pack_bf16x2(o[d][4 * rr + 2] * inv + bf_lo(o1[d][rr].y), o[d][4 * rr + 3] * inv + bf_hi(o1[d][rr].y));
                *reinterpret_cast<uint2*>(op + d * 32 + rr * 8) = w;
            }
        }
    }
}

// ================================================================================================
// Building blocks shared with the software-pipelined kernel below (Q fragments are read from LDS there).
// A "ping-pong" variant (the two wave groups half a tile apart: one on the matrix pipe while its SIMD
// partner runs the softmax, two barriers per tile) was measured at 584 TFLOP/s against 1035-1078 for the
// lock-step kernel above -- its control flow made hipcc drain vmcnt at every half-step -- and removed.
// ================================================================================================
#define PP_QK(S_, slot_)                                                                          \
    {                                                                                             \
        _Pragma("unroll") for (int f = 0; f < 2; ++f)                                             \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) S_[f][e] = 0.f;                        \
        const u16* ks_ = Ks + ((slot_) * KVBLK + ql) * K_LD + g * 8;                              \
        _Pragma("unroll") for (int ks = 0; ks < HD / 16; ++ks) {                                  \
            const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(qs_ + ks * 16);                   \
            _Pragma("unroll") for (int f = 0; f < 2; ++f) {                                       \
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks_ + f * 32 * K_LD + ks * 16); \
                if (SWP_ABL & 1) { /* timing ablation: every K fragment read twice */             \
                    const bf16x8 kd_ = *reinterpret_cast<const bf16x8*>(ks_ + (f ^ 1) * 32 * K_LD + ks * 16 + 8 * K_LD); \
                    asm volatile("" :: "v"(kd_));                                                 \
                }                                                                                 \
                S_[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfr, S_[f], 0, 0, 0);         \
            }                                                                                     \
        }                                                                                         \
    }
#define PP_PV(slot_)                                                                              \
    {                                                                                             \
        const u16* vs_ = Vs + ((slot_) * HD + ql) * V_LD + g * 8;                                 \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                        \
            _Pragma("unroll") for (int d = 0; d < HD / 32; ++d) {                                 \
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vs_ + d * 32 * V_LD + ks * 16); \
                if (SWP_ABL & 1) { /* timing ablation: every V fragment read twice */             \
                    const bf16x8 vd_ = *reinterpret_cast<const bf16x8*>(vs_ + (d ^ 1) * 32 * V_LD + ks * 16 + 8 * V_LD); \
                    asm volatile("" :: "v"(vd_));                                                 \
                }                                                                                 \
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], o[d], 0, 0, 0);        \
            }                                                                                     \
        }                                                                                         \
    }
#define PP_LOAD_K(tt_)                                                                                    \
    {                                                                                                     \
        const int seg_ = (tt_) / tps, key0_ = ((tt_) - seg_ * tps) * KVBLK;                               \
        const u16* kp_ = kbase + (int64_t)seg_ * p.k_ss;                                                  \
        kr0 = *reinterpret_cast<const uint4*>(kp_ + (int64_t)min(key0_ + krow, p.Lk - 1) * p.k_rs);       \
        kr1 = *reinterpret_cast<const uint4*>(kp_ + (int64_t)min(key0_ + krow + 32, p.Lk - 1) * p.k_rs);  \
    }
#define PP_LOAD_V(tt_)                                                                                    \
    {                                                                                                     \
        const int seg_ = (tt_) / tps, key0_ = ((tt_) - seg_ * tps) * KVBLK;                               \
        const u16* vp_ = vbase + (int64_t)seg_ * p.vt_ss + key0_;                                         \
        vr0 = *reinterpret_cast<const uint4*>(vp_ + (int64_t)vrow * p.Lkp);                               \
        vr1 = *reinterpret_cast<const uint4*>(vp_ + (int64_t)(vrow + 64) * p.Lkp);                        \
    }
#define PP_STORE_K(slot_)                                                                          \
    {                                                                                              \
        *reinterpret_cast<uint4*>(Ks + ((slot_) * KVBLK + krow) * K_LD + kcc * 8) = kr0;           \
        *reinterpret_cast<uint4*>(Ks + ((slot_) * KVBLK + krow + 32) * K_LD + kcc * 8) = kr1;      \
    }
#define PP_STORE_V(slot_)                                                                          \
    {                                                                                              \
        *reinterpret_cast<uint4*>(Vs + ((slot_) * HD + vrow) * V_LD + vcc * 8) = vr0;              \
        *reinterpret_cast<uint4*>(Vs + ((slot_) * HD + vrow + 64) * V_LD + vcc * 8) = vr1;         \
    }


// ================================================================================================
// Software-pipelined structure.  One code path for all 8 waves; inside a wave the QK^T MFMAs of tile
// t+1 are issued interleaved (sched_group_barrier) with the softmax VALU work of tile t, which is
// independent of them -- the matrix pipe runs asynchronously, so the VALU instructions placed in the
// gaps between MFMA issues are free.  Then P(t).V(t).  Straight-line loop body (tile indices past the
// end are clamped instead of branched around, so hipcc's s_waitcnt placement stays exact).
//   LDS: K ring 2 slots (K(t+1) read in iteration t, K(t+2) written at its end), V ring 2 slots,
//   Q rows of the block (re-read per tile; frees 32 VGPRs for the second score tile).
// ================================================================================================
// row max of a 64-key score tile held as S_[2][16] per lane, combined across the row's two lanes
// (lane, lane^32) with v_permlane32_swap: r[0] = {lo, lo}, r[1] = {hi, hi} -> max(r[0], r[1])
#define SWP_ROWMAX(S_, OUT_)                                                                      \
    {                                                                                             \
        float mx_ = S_[0][0];                                                                     \
        _Pragma("unroll") for (int f = 0; f < 2; ++f)                                             \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) mx_ = fmaxf(mx_, S_[f][r]);            \
        const unsigned mi_ = __float_as_uint(mx_);                                                \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(mi_, mi_, false, false);                \
        OUT_ = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                           \
    }

// exp / sum / pack part of the online softmax, row max MX_ already known
#define SWP_SOFTMAX(S_, MX_)                                                                      \
    const float m_new_ = fmaxf(m_run, MX_);                                                       \
    const float alpha_ = __builtin_amdgcn_exp2f((m_run - m_new_) * sl2);                          \
    const float msc_ = m_new_ * sl2;                                                              \
    float rs_ = 0.f;                                                                              \
    _Pragma("unroll") for (int f = 0; f < 2; ++f)                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                          \
            const float pv_ = __builtin_amdgcn_exp2f(S_[f][r] * sl2 - msc_);                      \
            S_[f][r] = pv_;                                                                       \
            rs_ += pv_;                                                                           \
        }                                                                                         \
    l_run = l_run * alpha_ + rs_;                                                                 \
    const bool moved_ = !__all(m_new_ == m_run);                                                  \
    m_run = m_new_;                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                            \
        uint4 u_;                                                                                 \
        const int f = ks >> 1, r0 = (ks & 1) * 8;                                                 \
        u_.x = pack_bf16x2(S_[f][r0 + 0], S_[f][r0 + 1]);                                         \
        u_.y = pack_bf16x2(S_[f][r0 + 2], S_[f][r0 + 3]);                                         \
        u_.z = pack_bf16x2(S_[f][r0 + 4], S_[f][r0 + 5]);                                         \
        u_.w = pack_bf16x2(S_[f][r0 + 6], S_[f][r0 + 7]);                                         \
        pf[ks] = __builtin_bit_cast(bf16x8, u_);                                                  \
    }

#define SWP_MASK(S_, T_)                                                                          \
    if (tail < KVBLK && ((T_) % tps) == tps - 1) {                                                \
        _Pragma("unroll") for (int f = 0; f < 2; ++f)                                             \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                      \
                const int key = 32 * f + (r & 3) + 8 * (r >> 2) + 4 * g;                          \
                if (key >= tail) S_[f][r] = -INFINITY;                                            \
            }                                                                                     \
    }

// MFMA / VALU(+TRANS) interleave groups: one MFMA, then n VALU instructions
#define SWP_GA __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, GA, 0);
#define SWP_GB __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, GB, 0);
#define SWP_X16(G_) G_ G_ G_ G_ G_ G_ G_ G_ G_ G_ G_ G_ G_ G_ G_ G_
#define SWP_X4(G_) G_ G_ G_ G_
// block B with the four ds_write_b128 of the staged K / V^T tiles named in the group pipeline (WS = 1): hipcc then issues them
// one per MFMA gap at the end of the block instead of clustered 1-1-2 before the barrier (+0.7 % on three boxes; asking for
// them earlier stalls on vmcnt: the tile's global loads were issued at the top of the same iteration)
#define SWP_GBW __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, GB, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);

// keeps a value computed BEFORE this point (machine sinking would otherwise move the exp2 chain below
// the rescale branch, out of the block that holds the QK^T MFMAs it is meant to hide under)
#define SWP_PIN(x_) { const uint4 pin_ = __builtin_bit_cast(uint4, x_); asm volatile("" :: "v"(pin_.x), "v"(pin_.y), "v"(pin_.z), "v"(pin_.w)); }

// One pipelined iteration.  In: SC_ = scores of tile T_ (masked), MXC_ = their row max.
//   block A:  SN_ = QK^T(T_+1)          ||  exp / sum / pack of SC_ -> P fragments
//   (rare)    O *= alpha when some row max moved;  mask SN_ when T_+1 is a segment's ragged tile
//   block B:  O += V^T(T_) P^T(T_)      ||  MXN_ = row max of SN_
#define SWP_ITER(SC_, SN_, MXC_, MXN_, T_, HAS_NEXT_)                                              \
    {                                                                                              \
        const int cur_ = (T_) & 1;                                                                 \
        PP_LOAD_K(min((T_) + 2, ntiles - 1))                                                       \
        PP_LOAD_V(min((T_) + 1, ntiles - 1))                                                       \
        if (HAS_NEXT_) PP_QK(SN_, cur_ ^ 1)                                                        \
        SWP_SOFTMAX(SC_, MXC_)                                                                     \
        if (HAS_NEXT_) { SWP_X16(SWP_GA) }                                                         \
        SWP_PIN(pf[0]) SWP_PIN(pf[1]) SWP_PIN(pf[2]) SWP_PIN(pf[3])                                \
        if (moved_) {                                                                              \
            _Pragma("unroll") for (int d = 0; d < HD / 32; ++d)                                    \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) o[d][e] *= alpha_;                  \
        }                                                                                          \
        if (HAS_NEXT_) SWP_MASK(SN_, (T_) + 1)                                                     \
        PP_PV(cur_)                                                                                \
        if (HAS_NEXT_) {                                                                           \
            SWP_ROWMAX(SN_, MXN_)                                                                  \
            if (WS == 1) { SWP_X4(SWP_GB) SWP_X4(SWP_GB) SWP_X4(SWP_GBW) SWP_X4(SWP_GB) }          \
            else { SWP_X16(SWP_GB) }                                                               \
        }                                                                                          \
        PP_STORE_K(cur_)                                                                           \
        PP_STORE_V(cur_ ^ 1)                                                                       \
        __syncthreads();                                                                           \
    }

template <int GA, int GB, int SWP_ABL = 0, int WS = 0>   // VALU(+TRANS) instructions scheduled into each MFMA gap of block A / block B
__global__ __launch_bounds__(ATT_THREADS, 2) void flash_attn_swp_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* Ks = smem;                              // [2][KVBLK][K_LD]
    u16* Vs = smem + 2 * KVBLK * K_LD;           // [2][HD][V_LD]
    u16* Qs = Vs + 2 * HD * V_LD;                // [8 waves][32][K_LD]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, g = lane >> 5;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int q0 = blockIdx.x * QBLK + wave * 32;
    const unsigned long long clk0 = p.probe ? __builtin_amdgcn_s_memtime() : 0ull;

    const u16* qs_ = Qs + (wave * 32 + ql) * K_LD + g * 8;
    {
        const int qrow = min(q0 + ql, p.Lq - 1);
        const u16* qp = p.q + b * p.q_bs + (int64_t)qrow * p.q_rs + (int64_t)h * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks)
            *reinterpret_cast<uint4*>(const_cast<u16*>(qs_) + ks * 16) = *reinterpret_cast<const uint4*>(qp + ks * 16);
    }
    const int krow = tid >> 4, kcc = tid & 15;
    const int vrow = tid >> 3, vcc = tid & 7;
    const u16* kbase = p.k + b * p.k_bs + (int64_t)h * HD + kcc * 8;
    const u16* vbase = p.vt + b * p.vt_bs + (int64_t)h * HD * p.Lkp + vcc * 8;
    const int tps = p.Lkp / KVBLK;
    const int ntiles = tps * p.n_seg;
    const int tail = p.Lk - (tps - 1) * KVBLK;
    const float sl2 = p.sl2;
    uint4 kr0, kr1, vr0, vr1;

    f32x16 o[HD / 32];
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 sa[2], sb[2];
    bf16x8 pf[4];

    // prologue: K(0) -> slot 0, K(1) -> slot 1, V(0) -> slot 0; scores of tile 0
    PP_LOAD_K(0)
    PP_LOAD_V(0)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    PP_STORE_K(0)
    PP_STORE_V(0)
    PP_LOAD_K(min(1, ntiles - 1))
    __builtin_amdgcn_s_waitcnt(0x0F70);
    PP_STORE_K(1)
    __syncthreads();
    float mxa, mxb = 0.f;
    PP_QK(sa, 0)
    SWP_MASK(sa, 0)
    SWP_ROWMAX(sa, mxa)
    __syncthreads();   // slot 0 is rewritten with K(2) at the end of iteration 0

    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
        SWP_ITER(sa, sb, mxa, mxb, t, true)
        SWP_ITER(sb, sa, mxb, mxa, t + 1, true)
    }
    if (ntiles - t == 2) {
        SWP_ITER(sa, sb, mxa, mxb, t, true)
        SWP_ITER(sb, sa, mxb, mxa, t + 1, false)
    } else {
        SWP_ITER(sa, sb, mxa, mxb, t, false)
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + ql;
    if (qrow < p.Lq) {
        u16* op = p.o + b * p.o_bs + (int64_t)qrow * p.o_rs + (int64_t)h * HD + 4 * g;
#pragma unroll
        for (int d = 0; d < HD / 32; ++d) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[d][4 * rr + e] * inv;
                uint2* dst = reinterpret_cast<uint2*>(op + d * 32 + rr * 8);
                if (p.accumulate) {
                    const uint2 old = *dst;
                    v[0] += bf_lo(old.x); v[1] += bf_hi(old.x);
                    v[2] += bf_lo(old.y); v[3] += bf_hi(old.y);
                }
                uint2 w;
                w.x = pack_bf16x2(v[0], v[1]);
                w.y = pack_bf16x2(v[2], v[3]);
                *dst = w;
            }
        }
    }
    if (p.probe && tid == 0) {
        atomicAdd(&g_attn_clk[0], (unsigned long long)__builtin_amdgcn_s_memtime() - clk0);
        atomicAdd(&g_attn_clk[1], 1ull);
    }
}

// Kernel selection (A/B measured on one box, B=2 x 8 heads x 48 832 keys, random data; lock-step + rescale
// skip = 100 %):  bit 0 s_setprio around MFMA clusters -3 %;  bit 1 rescale skip +4 % (in the baseline);
// bit 6 4-wave workgroups x 2 per CU -7 %;  bit 8 LDS-DMA K/V staging -3 %;  bit 3 software-pipelined kernel
// with (bits 12+) 9/3 VALU per MFMA gap -3 %, 6/4 -4 %, 5/5 +3 %, **4/4 +4 % (default)**, 3/3 +2.5 %.
// bits 4/5 are ablations (wrong results): no softmax -14 % time, no staging/barrier -15 %, neither -35 %.

// ================================================================================================
// Removed after measurement (kept in the history and in DESIGN.md section 4.2): a half-tile (32-key) pipeline with Q in
// registers -- 8 waves x 32 rows (-5 %) and 4 waves x 64 rows, one wave per SIMD with 512 registers, register / LDS-DMA
// staging, P.V as inline asm with AGPR accumulators (610-850 TFLOP/s against 1120 here: with one wave per SIMD every VALU
// dependency of the softmax is exposed).
// ================================================================================================

// ================================================================================================
// attn4: the hand-scheduled 4-wave kernel (one wave per SIMD, 64 query rows per wave, O / Q / K fragments in the accumulator
// file, LDS-DMA rings, lazy rescale).  It is written as gfx950 assembly GENERATED by scail_amd/asmgen/attn4.py (csrc/attn4.s),
// assembled by build.py into a code object whose bytes are embedded here and loaded with hipModuleLoadData on first use.
// Kernel argument block = asmgen/attn4.py KERNARG_FMT.
// ================================================================================================
static const unsigned char k_attn4_hsaco[] = {
#include "attn4_hsaco.inc"
};
struct Attn4Args {
    const void* q; const void* k; const void* vt; void* o;
    int64_t q_bs, q_rs, k_ss, k_bs, k_rs, vt_ss, vt_bs, o_bs, o_rs;
    int32_t heads, Lq, Lk, Lkp, n_seg;
    float sl2, thr;
    int32_t nqb;                 // query blocks of 256 rows (192 for the _q3 kernel)
    uint32_t magic_nqb, magic_heads;   // ceil(2^31 / d): x / d == (2 x * magic) >> 32 for the workgroup-id decode
    int32_t xcd_mode;            // 1: ids congruent mod 8 (one XCD) share (batch, head) pairs -> K / V^T reuse in that XCD's L2;
                                 // 2 (any pair count): XCD x walks the x-th of 8 equal runs of the pair-major (pair, query block) items
    int32_t items_per_xcd;       // xcd_mode 2: ceil(n_items / 8); the grid is 8 * items_per_xcd
    int32_t n_items, item0;      // the launch covers items [item0, item0 + n_items) of the pair-major list (item0 == 0 in xcd_mode 1)
    uint32_t* restarts;          // optional device counter: + 1 per workgroup that restarts after an overflow of the optimistic pass (NULL: none)
};
static_assert(sizeof(Attn4Args) == 168, "Attn4Args must match asmgen/attn4.py KERNARG_SIZE");
// Code objects, kernel handles (and the GEMM's tile-order tables, gemm.hip) belong to ONE device: they are cached per HIP device
// id, so a process that drives several GPUs (a DiT on cuda:0 and another engine on cuda:1, a threaded multi-GPU host) launches the
// module loaded on the device that is current at the call.
static std::map<int, hipModule_t> g_attn4_modules;                // device -> loaded code object
static const char* const k_attn4_default = "scail_attn4_m16f";   // the shipped kernel
static std::string g_attn4_name = k_attn4_default;               // A/B variants of the measurement build replace it ("attn4_kernel:<suffix>")
static std::map<std::pair<int, std::string>, hipFunction_t> g_attn4_fns;
static std::mutex g_attn4_mutex;
static std::atomic<float> g_attn4_thr_log2{8.0f};   // lazy-rescale threshold: P <= 2^thr
static std::atomic<int> g_attn4_xcd{1};      // XCD-aware workgroup-id decode (A/B knob "attn4_xcd")

static int attn4_function(const std::string& name, hipFunction_t* fn) {
    std::lock_guard<std::mutex> lk(g_attn4_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        scail_set_error("attn4: hipGetDevice failed");
        return 2;
    }
    auto mit = g_attn4_modules.find(dev);
    if (mit == g_attn4_modules.end()) {
        hipModule_t mod = nullptr;
        hipError_t e = hipModuleLoadData(&mod, k_attn4_hsaco);
        if (e != hipSuccess) {
            scail_set_error(std::string("attn4: hipModuleLoadData failed: ") + hipGetErrorString(e));
            return 2;
        }
        mit = g_attn4_modules.emplace(dev, mod).first;
    }
    auto it = g_attn4_fns.find(std::make_pair(dev, name));
    if (it == g_attn4_fns.end()) {
        hipFunction_t f;
        hipError_t e = hipModuleGetFunction(&f, mit->second, name.c_str());
        if (e != hipSuccess) {
            scail_set_error("attn4: kernel " + name + " is not in the embedded code object: " + hipGetErrorString(e));
            return 2;
        }
        it = g_attn4_fns.emplace(std::make_pair(dev, name), f).first;
    }
    *fn = it->second;
    return 0;
}

// load the embedded code object and resolve the shipped kernel now (scail_dit_create calls this: a first launch inside
// hipStreamBeginCapture must not have to load a module)
static int attn4_cu_count();
int scail_attn4_preload() {
    hipFunction_t fn;
    // every shipped kernel the launch plan / the cross-attention dispatch can pick, and the per-device CU count the plan reads
    for (const char* name : {"scail_attn4_m16f", "scail_attn4_m16f_q3", "scail_attn4_x2"})
        if (int rc = attn4_function(name, &fn)) return rc;
    return attn4_cu_count() > 0 ? 0 : 2;
}

static std::atomic<int> g_attn4_mode{1};     // 1 = use attn4 where eligible (default), 0 = never (8-wave kernels only)
// option "cross4": scail_attn4_x2 for the two-set cross attention.  2 (default) = by key count: measured per 256-row item at the config-2
// query shape (profiles/r05_cross_x2_phase_probe.log): scail_attn4_x2 22.6 us + 1.5 us per key tile (1.2-1.3 in the hot loop, which a set
// reaches from 10 tiles on), cross_attn2_kernel 8.5 us + 2.17 us per tile -> the generated kernel wins from 21 tiles of both sets together
// (1024 + 64 keys: 2.67 vs 2.74 ms; 4096 + 64: 6.39 vs 9.04), the hipcc kernel below -- the SHIPPED shape (512 + 257 keys = 13 tiles:
// 2.55 vs 2.19 ms) stays on the hipcc kernel.  1 = wherever eligible, 0 = never.  DESIGN.md section 4.3 has the phase budget.
static std::atomic<int> g_cross4{2};
constexpr int64_t k_cross4_min_tiles = 21;
// Option state is read by every launching thread and may be set by another one: plain loads / stores of std::atomic<int> (relaxed is
// enough: an option is a hint about FUTURE launches, a launch reads each option once).  include/scail_hip.h scail_set_option.
static std::atomic<int> g_attn4_rows{0};       // query rows per workgroup: 0 = planned per launch (below), 256 / 192 = one height for every launch
static std::atomic<int> g_attn4_cus{0};        // option "attn4_cus": CUs the launch plan may count on (0 = all CUs of the device)
static thread_local int g_attn4_rows_hint = 0;   // set by a caller that knows more than one call can (scail_attn4_rows_hint)
static thread_local uint32_t* g_attn4_restart_ctr = nullptr;   // scail_flash_attn_count_restarts (include/scail_hip.h)
extern "C" int scail_flash_attn_count_restarts(uint32_t* device_counter) {
    SCAIL_REQUIRE((reinterpret_cast<uintptr_t>(device_counter) & 3) == 0, "the restart counter must be 4-byte aligned");
    g_attn4_restart_ctr = device_counter;
    return 0;
}
// ---- launch shape of one attention (round 5) ----------------------------------------------------------------------------------
// One workgroup occupies a CU (512 registers per lane, one wave per SIMD), so W workgroups take ceil(W / CUs) rounds of one tile each, and
// a launch whose last round is mostly empty wastes it: a sequence-parallel rank's launch (Ulysses, 8 ranks: 5 heads x 191 tiles of 256
// rows = 3.73 rounds) pays 4.  scail_attn4_m16f_q3 (192 rows: 3 of the 4 query blocks per wave, 102 of 136 MFMAs per key tile beside the
// same K / V^T traffic) costs 0.79 of a 256-row tile (measured: 0.780-0.791 at 5 / 10 / 40 / 80 pairs, profiles/r05_attn_sp_shape_probe.log).
// The plan: WHOLE rounds of 256-row tiles first, the remaining rows -- from a 768-row boundary on, so that both tilings agree on it -- as
// 192-row tiles in a second launch on the same stream (8 ranks: 3 rounds + 250 tiles of 192 rows = 3.79 round-equivalents instead of 4;
// 4 ranks: 6 + 2 x 0.79 = 7.58 instead of 8); a pure 192-row launch where that is cheaper; the single 256-row launch otherwise (one GPU:
// 59.7 rounds, nothing to gain).  Results do not depend on the plan except in workgroups that restart after an exp2 overflow of the
// optimistic pass (a key > 167 log2 units above the first tile's maximum): their rows are recomputed by the lazy-maximum loop, whose
// rounding differs in the last bit, and which rows share a workgroup depends on the tile height.
static const double k_attn4_q3_cost = 0.79;
struct Attn4Launch {
    int rows;                   // 256 | 192
    int64_t item0, n_items;     // items [item0, item0 + n_items) of the pair-major (pair, query tile) list of THIS tile height
};
// CUs of the current device (cached per device); 0 + scail_last_error when the runtime cannot say -- the plan does not guess
static int attn4_cu_count() {
    static std::map<int, int> cus;
    std::lock_guard<std::mutex> lk(g_attn4_mutex);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        scail_set_error("attn4: hipGetDevice failed");
        return 0;
    }
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) {
        scail_set_error("attn4: the device's compute-unit count is not available (hipDeviceAttributeMultiprocessorCount)");
        return 0;
    }
    return cus[dev] = n;
}
// Returns the number of launches (1 or 2), or 0 + scail_last_error when the device's CU count is unavailable.
// Constants of the plan (measured, profiles/r05_attn_sp_shape_probe.log): k_attn4_q3_cost = 0.79 (a 192-row tile against a 256-row one),
// a 1.5 % minimum gain before anything but the single 256-row launch is chosen, 0.02 round-equivalents for the gap between two launches.
// CUs: the device's count, or option "attn4_cus" when a concurrent kernel (RCCL's channels on a sequence-parallel rank) holds some.
static int attn4_plan(int64_t n_batch, int64_t heads, int64_t Lq, Attn4Launch out[2]) {
    const int64_t P = heads * n_batch, n4 = (Lq + 255) / 256, n3 = (Lq + 191) / 192, W4 = P * n4, W3 = P * n3;
    // an explicit option wins; the caller's hint (thread-local: two attentions side by side) applies to the automatic choice only
    const int opt = g_attn4_rows.load(std::memory_order_relaxed);
    const int forced = opt ? opt : g_attn4_rows_hint;
    if (forced == 192) { out[0] = {192, 0, W3}; return 1; }
    if (forced == 256) { out[0] = {256, 0, W4}; return 1; }
    int64_t cus = attn4_cu_count();
    if (cus <= 0) return 0;
    const int avail = g_attn4_cus.load(std::memory_order_relaxed);
    if (avail > 0 && avail < cus) cus = avail;
    auto rounds = [&](int64_t w) { return (double)((w + cus - 1) / cus); };
    double best = rounds(W4);
    out[0] = {256, 0, W4};
    int n = 1;
    if (rounds(W3) * k_attn4_q3_cost < best * 0.985) {
        best = rounds(W3) * k_attn4_q3_cost;
        out[0] = {192, 0, W3};
    }
    const double gap = 0.02;                               // the second launch starts when the first has drained: ~20 us of a ~1 ms round
    for (int64_t kk = W4 / cus; kk >= 1 && kk >= W4 / cus - 4; --kk) {
        const int64_t amax = kk * cus, p = amax / n4, c = (amax % n4) / 3;     // split at row 768 c of pair p
        if (p >= P) continue;
        const int64_t a = p * n4 + 3 * c, i3 = p * n3 + 4 * c, b = W3 - i3;
        if (a <= 0 || b <= 0) continue;
        const double cost = rounds(a) + rounds(b) * k_attn4_q3_cost + gap;
        if (cost < best * 0.985) {
            best = cost;
            out[0] = {256, 0, a};
            out[1] = {192, i3, b};
            n = 2;
        }
    }
    return n;
}
// A caller that issues several attention launches concurrently (the sequence-parallel executor's two side streams, csrc/dit_step.hip) fills
// the partial rounds with the other stream's workgroups: it pins the single 256-row launch for its calls on this thread.  0 = no hint.
void scail_attn4_rows_hint(int rows) { g_attn4_rows_hint = rows; }
// scail_attn4_m16f takes ANY key count >= 512 (ragged last tile: K rows fetched from 64 rows earlier, scores masked) and any scale (q in
// log2 units as it is, a raw scale through a one-time multiplication of the Q fragments in its prologue); what is left to the 8-wave
// kernel: short key sets, accumulate, slices beyond 32-bit byte offsets.  (A non-ragged A/B variant of the measurement build needs
// whole 64-key tiles.)
static bool attn4_variant_is(const char* what) { return g_attn4_name.find(what) != std::string::npos; }
static bool attn4_eligible(int64_t q_rs, int64_t k_rs, int64_t o_rs, int64_t Lq, int64_t Lk, int accumulate, bool prescaled) {
    const int64_t lim = (1ll << 30);     // elements -> 2^31 bytes
    const bool is_default = g_attn4_name == k_attn4_default;
    const bool ragged_ok = is_default, raw_ok = is_default || !(attn4_variant_is("m16f") || attn4_variant_is("m16g"));
    return (Lk % 64 == 0 || ragged_ok) && (prescaled || raw_ok) && Lk >= 512 && accumulate == 0 && Lq * q_rs < lim && Lk * k_rs < lim &&
           Lq * o_rs < lim && 128 * (Lk + 63) < lim;
}
// The kernel decodes its 1-D workgroup id with reciprocal multiplications that are exact while id * divisor < 2^31
// (asmgen/attn4.py magic31): query blocks^2 * heads * batch and heads^2 * batch must stay below that; larger grids run the 8-wave kernel.
static bool attn4_grid_ok(int64_t n_batch, int64_t heads, int64_t Lq) {
    const int64_t nqb = (Lq + 191) / 192, lim = 1ll << 31;       // the finer of the two query tilings
    return heads < (1 << 15) && n_batch < (1 << 15) && nqb * heads * n_batch < lim / nqb && heads * n_batch < lim / heads;
}

extern "C" int scail_flash_attn_rows_for(int64_t n_batch, int64_t heads, int64_t Lq) {
    if (n_batch <= 0 || heads <= 0 || Lq <= 0) return 256;
    Attn4Launch pl[2];
    const int n = attn4_plan(n_batch, heads, Lq, pl);
    return n == 0 ? -1 : (n == 2 ? 448 : pl[0].rows);
}

extern "C" int scail_flash_attn_kernel_for(int64_t q_rs, int64_t k_rs, int64_t o_rs, int64_t Lq, int64_t Lk, int accumulate, int prescaled) {
    return (g_attn4_mode && attn4_eligible(q_rs, k_rs, o_rs, Lq, Lk, accumulate, prescaled != 0)) ? 4 : 8;
}

int scail_gemm4_enable(int on);   // gemm.hip
// ---- runtime options of the product library (include/scail_hip.h) -----------------------------------------------------
int scail_conv4_enable(int v);      // conv.hip
int scail_row_wave_enable(int v);   // rowops.hip
int scail_conv_direct_enable(int v);   // conv.hip
int scail_conv4_cont_enable(int v);    // conv.hip
int scail_conv4_resnorm_enable(int v); // conv.hip
int scail_conv_s2_enable(int v);       // conv.hip
extern "C" int scail_set_option(const char* name, int value) {
    const std::string k(name ? name : "");
    if (k == "attn4") { g_attn4_mode = value != 0; return 0; }              // 0: 8-wave kernel for every shape
    if (k == "cross4") {                                                    // 2: scail_attn4_x2 from 21 key tiles on; 1: wherever eligible; 0: never
        SCAIL_REQUIRE(value >= 0 && value <= 2, "cross4 must be 0, 1 or 2");
        g_cross4 = value;
        return 0;
    }
    if (k == "attn4_rows") {                                                // query rows per workgroup of attn4: 0 = per-launch choice
        SCAIL_REQUIRE(value == 0 || value == 192 || value == 256, "attn4_rows must be 0 (automatic), 192 or 256");
        g_attn4_rows = value;
        return 0;
    }
    if (k == "attn4_xcd") {                                                 // 0: plain workgroup-id decode (no XCD-aware K / V^T sharing)
        g_attn4_xcd = value != 0;
        return 0;
    }
    if (k == "attn4_cus") {                                                 // CUs the launch plan may count on; 0 = all CUs of the device
        SCAIL_REQUIRE(value >= 0 && value <= 4096, "attn4_cus must be in [0, 4096]");
        g_attn4_cus = value;
        return 0;
    }
    if (k == "gemm4") return scail_gemm4_enable(value);                     // 0: the kernels of csrc/gemm.hip for every shape
    if (k == "conv4") return scail_conv4_enable(value);                     // 0: the kernels of csrc/conv.hip for every convolution
    if (k == "row_wave") return scail_row_wave_enable(value);
    if (k == "conv_direct") return scail_conv_direct_enable(value);
    if (k == "conv4_cont") return scail_conv4_cont_enable(value);           // 1: tile-continuation variants of the generated 96-channel kernels         // 0: the gather kernel for the HBM-bound convolutions too               // 0: block-per-row LayerNorm / RMSNorm kernels for every width
    if (k == "conv_s2") return scail_conv_s2_enable(value);                 // 0: the gather kernel for Resample's stride-2 convolution; 1 / 2: conv_s2_kernel with one / two frames per workgroup
    if (k == "conv4_resnorm") return scail_conv4_resnorm_enable(value);     // 0: scail_conv3d_cl_resid_norm = the two separate calls for every shape
    if (k == "attn4_thr") {                                                 // lazy-rescale threshold of attn4: P <= 2^value
        SCAIL_REQUIRE(value >= 0 && value <= 64, "attn4_thr must be in [0, 64] (log2 units)");
        g_attn4_thr_log2 = (float)value;
        return 0;
    }
    scail_set_error(std::string("scail_set_option: unknown option ") + k);
    return 1;
}

#ifdef SCAIL_ABLATIONS
// ---- measurement build only (include/scail_hip_ablation.h): A/B of kernel schedules, timing ablations, cycle probes ----------
int scail_gemm_tune(int v);
int scail_gemm4_knob(const char* knob, int value);
int scail_gemm_group_m(int v);
int scail_conv_tune(int v);
int scail_conv4_kernel(const char* suffix);
static int g_attn_variant = 8 | (2 << 12);
extern "C" int scail_tune_set(const char* knob, int value) {
    if (std::string(knob) == "attn_variant") {
        if (value & (512 | 1024 | 2048)) {
            scail_set_error("scail_tune_set: attn_variant " + std::to_string(value) + " selects a removed kernel (half-tile pipelines)");
            return 1;
        }
        g_attn_variant = value;
        return 0;
    }
    if (std::string(knob) == "attn4") { g_attn4_mode = value != 0; return 0; }              // 0: 8-wave kernels only
    if (std::string(knob) == "attn4_thr") { g_attn4_thr_log2 = (float)value; return 0; }    // lazy-rescale threshold (log2 units)
    if (std::string(knob) == "attn4_xcd") { g_attn4_xcd = value != 0; return 0; }
    if (std::string(knob).rfind("attn4_kernel", 0) == 0) {
        // A/B of the generated schedules: knob = "attn4_kernel" (default kernel) or "attn4_kernel:<suffix>" -> kernel
        // scail_attn4_<suffix> of the embedded code object (the variants exist only in the ablation build, SCAIL_ABLATIONS=1)
        std::string k(knob), name = k_attn4_default;
        if (k.size() > 13) name = "scail_attn4_" + k.substr(13);
        if (name == "scail_attn4_general") name = "scail_attn4";      // the 32x32x16 kernel (round 2's raw-scale path)
        g_attn4_name = name;
        hipFunction_t fn;
        return attn4_function(name, &fn);
    }
    if (std::string(knob).rfind("gemm4", 0) == 0) return scail_gemm4_knob(knob, value);     // "gemm4" on / off, "gemm4_kernel:<suffix>"
    if (std::string(knob) == "gemm_tile") return scail_gemm_tune(value);
    if (std::string(knob) == "conv_halo") return scail_conv_tune(value);                    // 0-5 halo layouts; 10 / 11: generated conv4 kernel off / on
    if (std::string(knob).rfind("conv4_kernel", 0) == 0) return scail_conv4_kernel(std::string(knob).size() > 13 ? knob + 12 : "");   // "conv4_kernel:_<suffix>"
    if (std::string(knob) == "gemm_group_m") return scail_gemm_group_m(value);
    scail_set_error(std::string("scail_tune_set: unknown knob ") + knob);
    return 1;
}

#endif  // SCAIL_ABLATIONS

extern "C" int scail_flash_attn_bf16(const scail_bf16* q, int64_t q_bs, int64_t q_rs,
                                     const scail_bf16* k, int64_t k_ss, int64_t k_bs, int64_t k_rs,
                                     const scail_bf16* vt, int64_t vt_ss, int64_t vt_bs,
                                     scail_bf16* o, int64_t o_bs, int64_t o_rs,
                                     int64_t n_batch, int64_t heads, int64_t Lq, int64_t Lk, int64_t n_seg,
                                     float scale, int accumulate, void* stream) {
    SCAIL_REQUIRE(Lk >= 1 && n_seg >= 1, "need at least one key");
    SCAIL_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && o_rs % 4 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 &&
                      k_ss % 8 == 0 && vt_ss % 8 == 0 && o_bs % 4 == 0,
                  "strides must keep 16-byte (q,k,vt) / 8-byte (o) alignment");
    SCAIL_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (reinterpret_cast<uintptr_t>(k) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(vt) & 15) == 0 && (reinterpret_cast<uintptr_t>(o) & 7) == 0,
                  "pointer alignment");
    const bool prescaled = scale == SCAIL_ATTN_Q_PRESCALED;        // q already carries scale * log2(e)
    SCAIL_REQUIRE(prescaled || (scale > 0.0f && scale < 3.0e38f), "scale must be a positive finite number or SCAIL_ATTN_Q_PRESCALED");
    const float sl2 = prescaled ? 1.0f : scale * 1.4426950408889634f;
    const int64_t Lkp = (Lk + 63) / 64 * 64;
    SCAIL_REQUIRE(vt_bs == 0 || vt_bs == heads * HD * Lkp, "vt batch stride must be 0 or heads*128*ceil64(Lk)");
    SCAIL_REQUIRE(Lq < (1ll << 31) && Lkp * n_seg < (1ll << 31), "sequence too long");
    if (Lq == 0 || n_batch == 0) return 0;
    static ScailDeviceOnce attr_set;
    if (attr_set.need()) {
        const void* fns[] = {reinterpret_cast<const void*>(&flash_attn_swp_kernel<4, 4, 0, 1>),
#ifdef SCAIL_ABLATIONS
                             reinterpret_cast<const void*>(&flash_attn_swp_kernel<5, 5>), reinterpret_cast<const void*>(&flash_attn_swp_kernel<4, 4>),
                             reinterpret_cast<const void*>(&flash_attn_swp_kernel<6, 4>), reinterpret_cast<const void*>(&flash_attn_swp_kernel<3, 3>),
                             reinterpret_cast<const void*>(&flash_attn_swp_kernel<9, 3>), reinterpret_cast<const void*>(&flash_attn_swp_kernel<4, 4, 1>),
#endif
        };
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_PP_LDS_BYTES);
            if (e != hipSuccess) {
                scail_set_error(std::string("flash_attn: hipFuncSetAttribute failed: ") + hipGetErrorString(e));
                return 2;
            }
        }
#ifdef SCAIL_ABLATIONS
        const void* lock[] = {reinterpret_cast<const void*>(&flash_attn_kernel<258, 8>), reinterpret_cast<const void*>(&flash_attn_kernel<2, 4>),
                              reinterpret_cast<const void*>(&flash_attn_kernel<0, 8>), reinterpret_cast<const void*>(&flash_attn_kernel<1, 8>),
                              reinterpret_cast<const void*>(&flash_attn_kernel<2, 8>), reinterpret_cast<const void*>(&flash_attn_kernel<3, 8>),
                              reinterpret_cast<const void*>(&flash_attn_kernel<18, 8>), reinterpret_cast<const void*>(&flash_attn_kernel<34, 8>),
                              reinterpret_cast<const void*>(&flash_attn_kernel<50, 8>)};
        for (const void* f : lock) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS_BYTES);
#endif
        attr_set.done();
    }
    if (g_attn4_mode && attn4_eligible(q_rs, k_rs, o_rs, Lq, Lk, accumulate, prescaled) && attn4_grid_ok(n_batch, heads, Lq)) {
        // scail_attn4_m16f (or the variant chosen with the measurement build's "attn4_kernel" knob).  The 16x16x32 "fold" kernels
        // work on scores in log2 units: sl2 = 0 says q carries scale * log2(e) already, any other value is multiplied into the Q
        // fragments once in the prologue; their lazy-rescale threshold is in log2 units.  The 32x32x16 variants of the
        // measurement build take sl2 per score and the threshold in raw-score units.
        hipFunction_t fn;
        const bool fold = attn4_variant_is("m16f") || attn4_variant_is("m16g");
        // launch plan (tile heights): the 192-row form exists for the shipped kernel only
        Attn4Launch plan[2];
        int n_launch = 1;
        if (g_attn4_name == k_attn4_default) {
            n_launch = attn4_plan(n_batch, heads, Lq, plan);
            if (n_launch == 0) return 2;
        } else {
            plan[0] = {256, 0, (Lq + 255) / 256 * heads * n_batch};
        }
        const int64_t pairs = heads * n_batch;
        for (int li = 0; li < n_launch; ++li) {
            const Attn4Launch& pl = plan[li];
            if (int rc = attn4_function(pl.rows == 192 ? g_attn4_name + "_q3" : g_attn4_name, &fn)) return rc;
            Attn4Args a;
            a.q = q; a.k = k; a.vt = vt; a.o = o;
            a.q_bs = q_bs; a.q_rs = q_rs; a.k_ss = k_ss; a.k_bs = k_bs; a.k_rs = k_rs; a.vt_ss = vt_ss; a.vt_bs = vt_bs;
            a.o_bs = o_bs; a.o_rs = o_rs;
            a.heads = (int32_t)heads; a.Lq = (int32_t)Lq; a.Lk = (int32_t)Lk; a.Lkp = (int32_t)Lkp; a.n_seg = (int32_t)n_seg;
            a.sl2 = fold ? (prescaled ? 0.0f : sl2) : sl2;
            const float thr_log2 = g_attn4_thr_log2;
            a.thr = fold ? thr_log2 : thr_log2 / sl2;
            a.nqb = (int32_t)((Lq + pl.rows - 1) / pl.rows);
            a.magic_nqb = (uint32_t)(((1ull << 31) + a.nqb - 1) / a.nqb);
            a.magic_heads = (uint32_t)(((1ull << 31) + heads - 1) / heads);
            a.n_items = (int32_t)pl.n_items;
            a.item0 = (int32_t)pl.item0;
            a.restarts = g_attn4_restart_ctr;
            a.items_per_xcd = (a.n_items + 7) / 8;
            // XCD-aware ids (measured, profiles/r05_attn_sp_shape_probe.log): fewer than 8 pairs -> plain decode (all XCDs stream the one
            // pair in step; 5 pairs: 4 % faster than a run per XCD); whole launches of 8 k pairs -> pairs dealt round-robin (mode 1);
            // anything else -> 8 equal runs of the item list (mode 2)
            a.xcd_mode = (!g_attn4_xcd || pairs < 8) ? 0 : ((pairs % 8 == 0 && n_launch == 1) ? 1 : 2);
            size_t sz = sizeof(a);
            void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            hipError_t e = hipModuleLaunchKernel(fn, (unsigned)(a.xcd_mode == 2 ? 8 * a.items_per_xcd : a.n_items), 1, 1, 256, 1, 1, 0,
                                                 (hipStream_t)stream, nullptr, extra);
            if (e != hipSuccess) {
                scail_set_error(std::string("attn4: launch failed: ") + hipGetErrorString(e));
                return 2;
            }
        }
        return 0;
    }
    AttnParams p;
    p.q = q; p.q_bs = q_bs; p.q_rs = q_rs;
    p.k = k; p.k_ss = k_ss; p.k_bs = k_bs; p.k_rs = k_rs;
    p.vt = vt; p.vt_ss = vt_ss; p.vt_bs = vt_bs;
    p.o = o; p.o_bs = o_bs; p.o_rs = o_rs;
    p.heads = (int)heads; p.Lq = (int)Lq; p.Lk = (int)Lk; p.Lkp = (int)Lkp; p.n_seg = (int)n_seg;
    p.sl2 = sl2;
    p.accumulate = accumulate;
    p.probe = 0;
    dim3 grid((unsigned)((Lq + QBLK - 1) / QBLK), (unsigned)heads, (unsigned)n_batch);
#ifdef SCAIL_ABLATIONS
    // A/B schedules of the 8-wave kernel family (all parity-tested; measurements in DESIGN.md section 4.2)
    p.probe = (g_attn_variant >> 20) & 1;
    if (g_attn_variant != (8 | (2 << 12))) {
        if (g_attn_variant & 8) {
            const int sub = (g_attn_variant >> 12) & 15;     // A/B of the interleave density
            static bool swp_attr = false;
            if (!swp_attr) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_swp_kernel<5, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, ATT_PP_LDS_BYTES);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_swp_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, ATT_PP_LDS_BYTES);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_swp_kernel<6, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, ATT_PP_LDS_BYTES);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_swp_kernel<3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, ATT_PP_LDS_BYTES);
    #ifdef SCAIL_ABLATIONS
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_swp_kernel<4, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, ATT_PP_LDS_BYTES);
    #endif
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_swp_kernel<4, 4, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, ATT_PP_LDS_BYTES);
                swp_attr = true;
            }
            if (sub == 1) hipLaunchKernelGGL((flash_attn_swp_kernel<5, 5>), grid, dim3(ATT_THREADS), ATT_PP_LDS_BYTES, (hipStream_t)stream, p);
            else if (sub == 2) hipLaunchKernelGGL((flash_attn_swp_kernel<4, 4, 0, 1>), grid, dim3(ATT_THREADS), ATT_PP_LDS_BYTES, (hipStream_t)stream, p);   // default
            else if (sub == 3) hipLaunchKernelGGL((flash_attn_swp_kernel<6, 4>), grid, dim3(ATT_THREADS), ATT_PP_LDS_BYTES, (hipStream_t)stream, p);
            else if (sub == 4) hipLaunchKernelGGL((flash_attn_swp_kernel<3, 3>), grid, dim3(ATT_THREADS), ATT_PP_LDS_BYTES, (hipStream_t)stream, p);
    #ifdef SCAIL_ABLATIONS
            else if (sub == 5) hipLaunchKernelGGL((flash_attn_swp_kernel<4, 4, 1>), grid, dim3(ATT_THREADS), ATT_PP_LDS_BYTES, (hipStream_t)stream, p);
    #endif
            else if (sub == 6) hipLaunchKernelGGL((flash_attn_swp_kernel<4, 4>), grid, dim3(ATT_THREADS), ATT_PP_LDS_BYTES, (hipStream_t)stream, p);   // 4/4 without the store placement
            else hipLaunchKernelGGL((flash_attn_swp_kernel<9, 3>), grid, dim3(ATT_THREADS), ATT_PP_LDS_BYTES, (hipStream_t)stream, p);
            return scail_check_launch("flash_attn");
        }
        if (g_attn_variant & 256) {  // LDS-DMA staging
            hipLaunchKernelGGL((flash_attn_kernel<258, 8>), grid, dim3(ATT_THREADS), ATT_LDS_BYTES, (hipStream_t)stream, p);
            return scail_check_launch("flash_attn");
        }
        if (g_attn_variant & 64) {   // 4-wave workgroups, two per CU
            dim3 grid4((unsigned)((Lq + 127) / 128), (unsigned)heads, (unsigned)n_batch);
            hipLaunchKernelGGL((flash_attn_kernel<2, 4>), grid4, dim3(256), ATT_LDS_BYTES, (hipStream_t)stream, p);
            return scail_check_launch("flash_attn");
        }
    #ifdef SCAIL_ABLATIONS
        if ((g_attn_variant & 0xFFFFF) == 18) { hipLaunchKernelGGL((flash_attn_kernel<18, 8>), grid, dim3(ATT_THREADS), ATT_LDS_BYTES, (hipStream_t)stream, p); return scail_check_launch("flash_attn"); }
        if ((g_attn_variant & 0xFFFFF) == 34) { hipLaunchKernelGGL((flash_attn_kernel<34, 8>), grid, dim3(ATT_THREADS), ATT_LDS_BYTES, (hipStream_t)stream, p); return scail_check_launch("flash_attn"); }
        if ((g_attn_variant & 0xFFFFF) == 50) { hipLaunchKernelGGL((flash_attn_kernel<50, 8>), grid, dim3(ATT_THREADS), ATT_LDS_BYTES, (hipStream_t)stream, p); return scail_check_launch("flash_attn"); }
    #endif
        switch (g_attn_variant & 3) {
            case 0: hipLaunchKernelGGL((flash_attn_kernel<0, 8>), grid, dim3(ATT_THREADS), ATT_LDS_BYTES, (hipStream_t)stream, p); break;
            case 1: hipLaunchKernelGGL((flash_attn_kernel<1, 8>), grid, dim3(ATT_THREADS), ATT_LDS_BYTES, (hipStream_t)stream, p); break;
            case 2: hipLaunchKernelGGL((flash_attn_kernel<2, 8>), grid, dim3(ATT_THREADS), ATT_LDS_BYTES, (hipStream_t)stream, p); break;
            default: hipLaunchKernelGGL((flash_attn_kernel<3, 8>), grid, dim3(ATT_THREADS), ATT_LDS_BYTES, (hipStream_t)stream, p); break;
        }
        return scail_check_launch("flash_attn");
    }
#endif
    // the 8-wave software-pipelined kernel: 4 / 4 VALU per MFMA gap, staging stores placed one per gap
    hipLaunchKernelGGL((flash_attn_swp_kernel<4, 4, 0, 1>), grid, dim3(ATT_THREADS), ATT_PP_LDS_BYTES, (hipStream_t)stream, p);
    return scail_check_launch("flash_attn");
}

// ---- cross attention over two key sets on the generated 4-wave pipeline (scail_attn4_x2) ----
struct Attn4X2Args {
    Attn4Args base;              // set 0 in the k / vt / k_bs / vt_bs / Lk / Lkp fields, n_seg = 1, xcd_mode 0, n_items = all items
    const void* k2; const void* vt2;
    int64_t k2_bs, vt2_bs;
    int32_t Lk2, Lkp2, n_wgs, pad;
};
static_assert(sizeof(Attn4X2Args) == 216, "Attn4X2Args must match asmgen/attn4.py X2_KERNARG_SIZE");
// one K row stride for both sets (the set switch keeps the K DMA lane offsets), at least one whole key tile per set (a ragged tile's
// missing rows are fetched from 64 rows earlier), 32-bit byte offsets inside a (batch, head) slice, exact reciprocal id decode
static bool cross4_eligible(int64_t q_rs, int64_t k1_rs, int64_t k2_rs, int64_t o_rs, int64_t Lq, int64_t Lk1, int64_t Lk2, int64_t n_batch, int64_t heads) {
    const int64_t lim = (1ll << 30);
    const bool want = g_cross4 == 1 || (g_cross4 == 2 && (Lk1 + 63) / 64 + (Lk2 + 63) / 64 >= k_cross4_min_tiles);
    return want && g_attn4_mode && k1_rs == k2_rs && Lk1 >= 64 && Lk2 >= 64 && Lq * q_rs < lim && Lq * o_rs < lim && Lk1 * k1_rs < lim &&
           Lk2 * k2_rs < lim && 128 * (Lk1 + 63) < lim && 128 * (Lk2 + 63) < lim && attn4_grid_ok(n_batch, heads, Lq);
}
extern "C" int scail_cross_attn2_kernel_for(int64_t q_rs, int64_t k1_rs, int64_t k2_rs, int64_t o_rs, int64_t Lq, int64_t Lk1, int64_t Lk2,
                                            int64_t n_batch, int64_t heads) {
    return cross4_eligible(q_rs, k1_rs, k2_rs, o_rs, Lq, Lk1, Lk2, n_batch, heads) ? 4 : 2;
}

extern "C" int scail_cross_attn2_bf16(const scail_bf16* q, int64_t q_bs, int64_t q_rs,
                                      const scail_bf16* k1, int64_t k1_bs, int64_t k1_rs, const scail_bf16* vt1, int64_t vt1_bs, int64_t Lk1,
                                      const scail_bf16* k2, int64_t k2_bs, int64_t k2_rs, const scail_bf16* vt2, int64_t vt2_bs, int64_t Lk2,
                                      scail_bf16* o, int64_t o_bs, int64_t o_rs,
                                      int64_t n_batch, int64_t heads, int64_t Lq, float scale, void* stream) {
    SCAIL_REQUIRE(Lk1 >= 1 && Lk2 >= 1 && Lk1 < (1ll << 30) && Lk2 < (1ll << 30), "cross_attn2: each key set needs at least one key");
    SCAIL_REQUIRE(heads >= 1 && heads < 65536 && n_batch < 65536 && Lq < (1ll << 31) - 256, "cross_attn2: grid limits");
    SCAIL_REQUIRE(q_rs % 8 == 0 && k1_rs % 8 == 0 && k2_rs % 8 == 0 && o_rs % 4 == 0 && q_bs % 8 == 0 && k1_bs % 8 == 0 && k2_bs % 8 == 0 &&
                      vt1_bs % 8 == 0 && vt2_bs % 8 == 0 && o_bs % 4 == 0,
                  "cross_attn2: strides must keep 16-byte (q, k, vt) / 8-byte (o) alignment");
    SCAIL_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (reinterpret_cast<uintptr_t>(k1) & 15) == 0 && (reinterpret_cast<uintptr_t>(k2) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(vt1) & 15) == 0 && (reinterpret_cast<uintptr_t>(vt2) & 15) == 0 && (reinterpret_cast<uintptr_t>(o) & 7) == 0,
                  "cross_attn2: pointer alignment");
    if (n_batch == 0 || Lq == 0) return 0;
    const bool prescaled = scale == SCAIL_ATTN_Q_PRESCALED;        // q already carries scale * log2(e)
    SCAIL_REQUIRE(prescaled || (scale > 0.0f && scale < 3.0e38f), "cross_attn2: scale must be a positive finite number or SCAIL_ATTN_Q_PRESCALED");
    if (cross4_eligible(q_rs, k1_rs, k2_rs, o_rs, Lq, Lk1, Lk2, n_batch, heads)) {
        // scail_attn4_x2 (csrc/attn4.s, asmgen/attn4.py Cfg.x2): the 4-wave pipeline of the self-attention kernel, persistent workgroups (one
        // per CU) walking over the (pair, 256-row query block) items, the key pipeline run once per key set and item
        hipFunction_t fn;
        if (int rc = attn4_function("scail_attn4_x2", &fn)) return rc;
        Attn4X2Args a;
        Attn4Args& b = a.base;
        const int64_t Lkp1 = (Lk1 + 63) / 64 * 64, Lkp2 = (Lk2 + 63) / 64 * 64;
        b.q = q; b.k = k1; b.vt = vt1; b.o = o;
        b.q_bs = q_bs; b.q_rs = q_rs; b.k_ss = 0; b.k_bs = k1_bs; b.k_rs = k1_rs; b.vt_ss = 0; b.vt_bs = vt1_bs; b.o_bs = o_bs; b.o_rs = o_rs;
        b.heads = (int32_t)heads; b.Lq = (int32_t)Lq; b.Lk = (int32_t)Lk1; b.Lkp = (int32_t)Lkp1; b.n_seg = 1;
        b.sl2 = prescaled ? 0.0f : scale * 1.4426950408889634f;
        b.thr = g_attn4_thr_log2;
        b.nqb = (int32_t)((Lq + 255) / 256);
        b.magic_nqb = (uint32_t)(((1ull << 31) + b.nqb - 1) / b.nqb);
        b.magic_heads = (uint32_t)(((1ull << 31) + heads - 1) / heads);
        b.xcd_mode = 0;
        b.n_items = (int32_t)(b.nqb * heads * n_batch);
        b.items_per_xcd = (b.n_items + 7) / 8;
        b.item0 = 0;
        b.restarts = nullptr;            // the cross attention is not counted (scail_flash_attn_count_restarts is about the self-attention)
        a.k2 = k2; a.vt2 = vt2; a.k2_bs = k2_bs; a.vt2_bs = vt2_bs; a.Lk2 = (int32_t)Lk2; a.Lkp2 = (int32_t)Lkp2;
        a.n_wgs = (int32_t)std::min<int64_t>(b.n_items, attn4_cu_count());
        a.pad = 0;
        size_t sz = sizeof(a);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        hipError_t e = hipModuleLaunchKernel(fn, (unsigned)a.n_wgs, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, extra);
        if (e != hipSuccess) {
            scail_set_error(std::string("cross_attn2 (attn4_x2): launch failed: ") + hipGetErrorString(e));
            return 2;
        }
        return 0;
    }
    static ScailDeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, X2_LDS_BYTES);
        if (e != hipSuccess) {
            scail_set_error(std::string("cross_attn2: hipFuncSetAttribute failed: ") + hipGetErrorString(e));
            return 2;
        }
        attr_set.done();
    }
    Cross2Params p;
    p.q = q; p.q_bs = q_bs; p.q_rs = q_rs;
    p.k0 = k1; p.k0_bs = k1_bs; p.k0_rs = k1_rs; p.vt0 = vt1; p.vt0_bs = vt1_bs; p.Lk0 = (int)Lk1; p.Lkp0 = (int)((Lk1 + KVBLK - 1) / KVBLK * KVBLK);
    p.k1 = k2; p.k1_bs = k2_bs; p.k1_rs = k2_rs; p.vt1 = vt2; p.vt1_bs = vt2_bs; p.Lk1 = (int)Lk2; p.Lkp1 = (int)((Lk2 + KVBLK - 1) / KVBLK * KVBLK);
    p.o = o; p.o_bs = o_bs; p.o_rs = o_rs;
    p.heads = (int)heads; p.Lq = (int)Lq;
    p.sl2 = prescaled ? 1.0f : scale * 1.44269504088896340736f;
    dim3 grid((unsigned)((Lq + 127) / 128), (unsigned)heads, (unsigned)n_batch);
    hipLaunchKernelGGL(cross_attn2_kernel, grid, dim3(X2_THREADS), X2_LDS_BYTES, (hipStream_t)stream, p);
    return scail_check_launch("cross_attn2");
}

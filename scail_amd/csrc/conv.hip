// Wan2.1 causal 3D VAE kernels (reference sgm/models/wan_vae.py) for gfx950.
//
// Layout: channels-last activations (T, H, W, C) bf16 -- a voxel's channels are contiguous, so every
// convolution is an implicit GEMM  y[voxel, cout] = sum_k A[voxel, k] . W[cout, k],  k = (tap, cin) with
// cin fastest, whose A rows are gathered 16 bytes (8 channels) at a time straight from the input tensor
// (zero for causal / spatial padding) into the same padded, double-buffered LDS tiles and MFMA 32x32x16
// main loop as scail_gemm_bf16.  One kernel covers CausalConv3d 3x3x3 / 1x1x1 / 3x1x1 (front-only time
// padding, wan_vae.py:17-36), the stride-2 temporal and spatial downsampling convs (:87-96) and the 3x3
// conv behind the nearest-exact 2x upsampling (:76-85; the upsample is folded into the gather: source
// pixel = (dst + tap - 1) >> 1).  The whole video is processed in one pass -- with 288 GB of HBM the
// reference's 1/4/4-frame chunking and its per-conv feature caches are unnecessary (the equivalence is
// pinned by oracle/wan_vae_oracle.py against the chunked reference).
#include "common.h"
#include <atomic>
#include <algorithm>
#include <map>
#include <mutex>

#define CBM 128
#define CBK 64
#define CLDT 72
#define CONV_THREADS 256

struct ConvParams {
    const u16* x;      // (Ti, Hi, Wi, Cin)
    const u16* w;      // (Cout_rows, Kpad), k = tap * Cin + c
    const float* bias; // (N) or null
    u16* y; int64_t ldc;          // output voxel v -> y + v * ldc
    const u16* resid; int64_t ldr;
    const float* gamma;           // EPI 4 (halo kernel, N <= tile width): RMS_norm gamma, output = SiLU(RMS_norm(conv)); conv_direct_kernel<.., true>: the consumer's gamma
    u16* y2 = nullptr;            // conv_direct_kernel<.., true>: SiLU(RMS_norm(bf16(conv + bias)) * gamma) beside the raw output (same ldc and frame mapping)
    int Ti, Hi, Wi, Cin;
    int To, Ho, Wo;               // output extents covered by this launch (M = To*Ho*Wo)
    int kt, kh, kw, st, sh, sw, pt, ph, pw, ups;
    int ot_mul, ot_off;           // output frame = to * ot_mul + ot_off
    int N, Kpad, Ktrue;
    int64_t M;
};

// BN_ / WM_ x WN_: 128 / 2 x 2 (wave tile 64 x 64) for wide layers, 96 / 4 x 1 (wave tile 32 x 96) for the
// 96-channel decoder / encoder stages, 64 / 4 x 1 (32 x 64) for narrow outputs -- no MFMA work on padding columns.
template <int EPI, int BN_, int WM_, int WN_>  // EPI 0: bias, 3: bias + residual
__global__ __launch_bounds__(CONV_THREADS) void conv_igemm_kernel(ConvParams p) {
    constexpr int WTM = CBM / WM_, WTN = BN_ / WN_, MI = WTM / 32, NI = WTN / 32, NWP = BN_ / 32;   // NWP: W staging passes
    static_assert(WM_ * WN_ == 4 && MI >= 1 && NI >= 1 && BN_ % 32 == 0 && NWP <= 4, "tile shape");
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* Xs = smem;
    u16* Ws = smem + 2 * CBM * CLDT;

    const int tiles_n = (p.N + BN_ - 1) / BN_;
    const int64_t pid_m = blockIdx.x / tiles_n;     // n fastest: the (few) n-tiles of one voxel block
    const int pid_n = blockIdx.x % tiles_n;         // run back to back and share the gathered input in L2
    const int64_t m0 = pid_m * CBM;
    const int n0 = pid_n * BN_;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN_, wn = wave % WN_;
    const int l31 = lane & 31, g = lane >> 5;
    const int srow = tid >> 3, kc = tid & 7;
    const int HWo = p.Ho * p.Wo;

    // ---- per-thread output rows (voxels) of the A tile: decode once ----
#define CROW(i_)                                                                         \
    int tb##i_, hb##i_, wb##i_;                                                          \
    {                                                                                    \
        const int64_t m_ = m0 + srow + 32 * i_;                                          \
        const bool ok_ = m_ < p.M;                                                       \
        const int64_t mm_ = ok_ ? m_ : 0;                                                \
        const int to_ = (int)(mm_ / HWo);                                                \
        const int r_ = (int)(mm_ - (int64_t)to_ * HWo);                                  \
        const int ho_ = r_ / p.Wo, wo_ = r_ - ho_ * p.Wo;                                \
        tb##i_ = ok_ ? to_ * p.st - p.pt : -(1 << 28);   /* invalid row -> always OOB */ \
        hb##i_ = ho_ * p.sh - p.ph;                                                      \
        wb##i_ = wo_ * p.sw - p.pw;                                                      \
    }                                                                                    \
    const u16* wptr##i_ = p.w + (int64_t)min(n0 + srow + 32 * i_, p.N - 1) * p.Kpad + kc * 8; \
    uint4 xr##i_, wr##i_ = make_uint4(0, 0, 0, 0);
    CROW(0) CROW(1) CROW(2) CROW(3)
    const int Hlim = p.ups ? 2 * p.Hi : p.Hi, Wlim = p.ups ? 2 * p.Wi : p.Wi;

    // this thread's k chunk (k = k0 + 8 kc) as (tap = (dt, dh, dw), channel c): advanced by 64 per k-tile
    // with carries instead of three integer divisions per tile
    int c_cur, dt_cur, dh_cur, dw_cur;
    {
        const int k_ = kc * 8;
        const int tap_ = k_ / p.Cin;
        c_cur = k_ - tap_ * p.Cin;
        const int khw = p.kh * p.kw;
        dt_cur = tap_ / khw;
        const int r2_ = tap_ - dt_cur * khw;
        dh_cur = r2_ / p.kw;
        dw_cur = r2_ - dh_cur * p.kw;
    }
#define CLOAD1(i_, k0_)                                                                          \
    {                                                                                            \
        const int ti_ = tb##i_ + dt_cur, hi_ = hb##i_ + dh_cur, wi_ = wb##i_ + dw_cur;            \
        const bool v_ = kvalid_ && ti_ >= 0 && ti_ < p.Ti && hi_ >= 0 && hi_ < Hlim && wi_ >= 0 && wi_ < Wlim; \
        const int hs_ = p.ups ? (hi_ >> 1) : hi_, ws_ = p.ups ? (wi_ >> 1) : wi_;                 \
        const int64_t off_ = (((int64_t)ti_ * p.Hi + hs_) * p.Wi + ws_) * p.Cin + c_cur;          \
        xr##i_ = v_ ? *reinterpret_cast<const uint4*>(p.x + off_) : make_uint4(0, 0, 0, 0);       \
        if (i_ < NWP) wr##i_ = *reinterpret_cast<const uint4*>(wptr##i_ + (k0_));                 \
    }
    // loads the chunk at the CURRENT tap state, then advances the state to the next k-tile
#define CLOAD(k0_)                                                         \
    {                                                                      \
        const bool kvalid_ = (k0_) + kc * 8 < p.Ktrue;                     \
        CLOAD1(0, k0_) CLOAD1(1, k0_) CLOAD1(2, k0_) CLOAD1(3, k0_)        \
        c_cur += CBK;                                                      \
        while (c_cur >= p.Cin) {                                           \
            c_cur -= p.Cin;                                                \
            if (++dw_cur == p.kw) {                                        \
                dw_cur = 0;                                                \
                if (++dh_cur == p.kh) { dh_cur = 0; ++dt_cur; }            \
            }                                                              \
        }                                                                  \
    }
#define CSTORE1(i_, buf_)                                                                         \
    *reinterpret_cast<uint4*>(Xs + ((buf_) * CBM + srow + 32 * i_) * CLDT + kc * 8) = xr##i_;      \
    if (i_ < NWP) *reinterpret_cast<uint4*>(Ws + ((buf_) * BN_ + srow + 32 * i_) * CLDT + kc * 8) = wr##i_;
#define CSTORE(buf_) CSTORE1(0, buf_) CSTORE1(1, buf_) CSTORE1(2, buf_) CSTORE1(3, buf_)

    f32x16 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int nk = p.Kpad / CBK;
    CLOAD(0)
    CSTORE(0)
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) CLOAD((t + 1) * CBK)
        const u16* xs = Xs + (cur * CBM + wm * WTM + l31) * CLDT + g * 8;
        const u16* ws = Ws + (cur * BN_ + wn * WTN + l31) * CLDT + g * 8;
#pragma unroll
        for (int ks = 0; ks < CBK / 16; ++ks) {
            bf16x8 wf[NI], xf[MI];
#pragma unroll
            for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(ws + i * 32 * CLDT + ks * 16);
#pragma unroll
            for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xs + i * 32 * CLDT + ks * 16);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
        if (t + 1 < nk) { CSTORE(cur ^ 1) }
        __syncthreads();
    }

#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int64_t m = m0 + wm * WTM + mi * 32 + l31;
        if (m >= p.M) continue;
        const int to = (int)(m / HWo);
        const int r = (int)(m - (int64_t)to * HWo);
        const int64_t vox = ((int64_t)(to * p.ot_mul + p.ot_off)) * HWo + r;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int n = n0 + wn * WTN + ni * 32 + 8 * rr + 4 * g;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * rr + e];
                if (p.bias != nullptr) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (EPI == 3) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(p.resid + vox * p.ldr + n);
                    v[0] += bf_lo(rv.x); v[1] += bf_hi(rv.x); v[2] += bf_lo(rv.y); v[3] += bf_hi(rv.y);
                }
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(p.y + vox * p.ldc + n) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Direct-gather kernel for the HBM-BOUND convolutions of the VAE (round 4): few k-steps, output or input traffic far above the MFMA
// time -- the stem CausalConv3d(3 -> 96) (K = 27 x 8: a 7.1 GB store), the temporal stride-2 convolution of downsample3d (3 x 1 x 1,
// K = 288), the 1 x 1 x 1 shortcuts (K = Cin).  The gather kernel above runs them latency-bound (0.7-2 TB/s: one k-tile in flight per
// workgroup, a barrier per k-tile, 8-byte partial-line stores).  Here:
//   * the whole weight matrix (N x Kpad, <= 64 KB) sits in LDS for the lifetime of a PERSISTENT workgroup of 8 waves;
//   * a wave owns 32 consecutive output voxels x all N channels and needs no barrier: the B operand of v_mfma_f32_32x32x16_bf16 is
//     8 consecutive k of one voxel = 8 channels of one tap = ONE 16-byte global load per lane and k-step, straight into the fragment
//     register (Cin % 8 == 0; padding = zero fill) -- every load of a tile is requested before the first MFMA, no staging through LDS;
//   * the output goes through a per-wave LDS strip (32 voxels x 96 channels) so that every store instruction writes whole 192-byte
//     voxel rows of neighbouring voxels (1 KB contiguous per wave instruction).
// KS = k-steps held in registers (16-byte quad each); NB = 32-channel blocks of the output (N = 32 NB <= 192).
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) uint4 g_conv_zero16 = {0u, 0u, 0u, 0u};     // what an out-of-range tap reads: the gathers need no branch and no zero fill
#define CD_THREADS 512
#define CD_STRIP_LD 104            // elements per staged voxel row: 96 channels + 8 (208 bytes: 16-byte aligned, conflict-light 8-byte writes)
// DUAL (round 6, NB = 3: the 96 channels of a voxel in one pass): the raw output AND, at y2, SiLU(RMS_norm(bf16(raw)) * gamma) -- the stem convolution of the
// encoder with the first ResidualBlock's input norm (the arithmetic of rms_silu_kernel on what the plain kernel stores)
template <int KS, int NB, bool DUAL = false>
__global__ __launch_bounds__(CD_THREADS, 1) void conv_direct_kernel(ConvParams p, int wld) {
    static_assert(!DUAL || NB == 3, "the norm needs every channel of a voxel in one strip pass");
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* Ws = smem;                                              // [N][wld]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u16* strip = smem + (int64_t)p.N * wld + wave * 32 * CD_STRIP_LD;      // [32][CD_STRIP_LD] of this wave
    float* Bs = reinterpret_cast<float*>(smem + (int64_t)p.N * wld + 8 * 32 * CD_STRIP_LD);     // [N] bias, then [N] gamma (DUAL): per-tile global loads of
    float* Gs = Bs + p.N;                                                                       // these 2 x 12 quads each waited for L1 / L2 (round 6)
    const int l31 = lane & 31, g = lane >> 5;
    const int ksteps = (p.Ktrue + 15) >> 4;                      // k-steps that hold real taps (<= KS)
    // ---- weights -> LDS, once (rows n, k < 16 ksteps; 16-byte chunks) ----
    {
        const int cpr = ksteps * 2;                              // chunks per row
        for (int c = tid; c < p.N * cpr; c += CD_THREADS) {
            const int n = c / cpr, kc = c - n * cpr;
            *reinterpret_cast<uint4*>(Ws + n * wld + kc * 8) = *reinterpret_cast<const uint4*>(p.w + (int64_t)n * p.Kpad + kc * 8);
        }
        for (int n = tid; n < p.N; n += CD_THREADS) {
            Bs[n] = p.bias != nullptr ? p.bias[n] : 0.f;
            if (DUAL) Gs[n] = p.gamma[n];
        }
    }
    __syncthreads();
    // ---- this lane's k chunks: chunk (2 ks + g) = 8 channels c0 .. c0 + 7 of tap (dt, dh, dw), packed c0 | dw << 16 | dh << 21 | dt << 26 ----
    int tapc[KS], offk[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = (2 * ks + g) * 8;
        const int tap = k / p.Cin;
        const int c0 = k - tap * p.Cin;
        const int khw = p.kh * p.kw;
        int dt = tap / khw;
        const int r2 = tap - dt * khw;
        const int dh = r2 / p.kw;
        const int dw = r2 - dh * p.kw;
        if (k >= p.Ktrue) dt = 31;                               // padding columns of the last k-step: a frame that never exists -> zeros
        tapc[ks] = c0 | (dw << 16) | (dh << 21) | (dt << 26);
        offk[ks] = ((dt * p.Hi + dh) * p.Wi + dw) * p.Cin + c0;     // element offset of the chunk from the voxel's tap (0, 0, 0), channel 0 (fits: frames < 2^31 bytes)
    }
    const int HWo = p.Ho * p.Wo;
    const int64_t ntile = (p.M + 31) / 32;
    // the fragments of a tile are dead once its MFMAs are issued: the NEXT tile's loads are requested before this tile's epilogue, whose
    // LDS round trip and stores then run under their latency
    // (frame, offset inside the frame) of a tile's first voxel are carried along instead of divided out: the 64-bit divisions by H * W -- one per lane and
    // tile here, one per 16-byte chunk in the store loop -- were a third of the stem's time (round 6).  A tile's 32 voxels cross at most one frame
    // boundary (H * W >= 32: checked by the dispatch).
    uint4 xr[KS];
    auto load_tile = [&](int64_t tile_, int tf_, int rem_) {
        const int64_t m = tile_ * 32 + l31;
        const bool ok = m < p.M;
        int to = tf_, r = rem_ + l31;
        if (r >= HWo) { r -= HWo; ++to; }
        const int ho = r / p.Wo, wo = r - ho * p.Wo;
        const int tb = ok ? to * p.st - p.pt : -(1 << 20), hb = ho * p.sh - p.ph, wb = wo * p.sw - p.pw;
        const u16* xb = p.x + (((int64_t)tb * p.Hi + hb) * p.Wi + wb) * p.Cin;      // the voxel's tap (0, 0, 0); one 64-bit address per tile, 32-bit offsets per chunk
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // every k-step loads: an out-of-range tap (padding, a k-step past Ktrue, a lane past M) reads the 16 zero bytes of g_conv_zero16 -- a select on the
            // address instead of a divergent branch and four zero-filling moves per k-step (round 6)
            const int tc = tapc[ks];
            const unsigned ti = (unsigned)(tb + (tc >> 26)), hi = (unsigned)(hb + ((tc >> 21) & 31)), wi = (unsigned)(wb + ((tc >> 16) & 31));
            const bool in = ks < ksteps && (tc >> 26) != 31 && ti < (unsigned)p.Ti && hi < (unsigned)p.Hi && wi < (unsigned)p.Wi;
            const uint4* src = in ? reinterpret_cast<const uint4*>(xb + offk[ks]) : &g_conv_zero16;
            xr[ks] = *src;
        }
    };
    const int64_t tstep = (int64_t)gridDim.x * 8;
    int64_t tile = (int64_t)blockIdx.x * 8 + wave;
    int tf = (int)((tile * 32) / HWo), rem = (int)(tile * 32 - (int64_t)tf * HWo);     // this tile's; (tfn, remn): the next tile's
    int tfn = tf, remn = rem;
    auto advance = [&]() {
        int64_t rr = (int64_t)remn + tstep * 32;
        while (rr >= HWo) { rr -= HWo; ++tfn; }
        remn = (int)rr;
    };
    if (tile < ntile) load_tile(tile, tf, rem);
    for (; tile < ntile; tile += tstep, tf = tfn, rem = remn) {
        f32x16 acc[NB];
        const u16* wrow = Ws + l31 * wld + g * 8;
        // W fragments two k-steps ahead of their MFMAs (hipcc otherwise issues every ds_read right in front of its MFMA and the LDS
        // latency of all KS x NB reads lines up on the critical path of a tile)
        constexpr int PF = KS * NB <= 42 ? 2 : 1;        // (the deeper form spills at KS x NB = 60 / 84 quads)
        bf16x8 wf[PF + 1][NB];
#pragma unroll
        for (int d = 0; d < PF; ++d)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wf[d][nb] = *reinterpret_cast<const bf16x8*>(wrow + nb * 32 * wld + (d < KS ? d : 0) * 16);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PF < KS) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wf[(ks + PF) % (PF + 1)][nb] = *reinterpret_cast<const bf16x8*>(wrow + nb * 32 * wld + (ks + PF) * 16);
            }
            if (ks == 0) {                       // (k-step 0 always exists: its MFMAs take the constant 0 as accumulator input -- no 16 x NB zeroing moves per tile)
                const bf16x8 xf = __builtin_bit_cast(bf16x8, xr[0]);
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][nb], xf, zero, 0, 0, 0);
            } else if (ks < ksteps) {
                const bf16x8 xf = __builtin_bit_cast(bf16x8, xr[ks]);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks % (PF + 1)][nb], xf, acc[nb], 0, 0, 0);
            }
        }
        advance();
        if (tile + tstep < ntile) load_tile(tile + tstep, tfn, remn);
        // ---- epilogue: 96 channels at a time through this wave's strip; lane -> (voxel l31, channels 32 nb + 8 rr + 4 g + e) ----
        const int64_t vox0 = tile * 32;
#pragma unroll
        for (int h = 0; h < (NB + 2) / 3; ++h) {
            const int nbs = (NB - 3 * h) < 3 ? (NB - 3 * h) : 3;                   // 32-channel blocks in this pass
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q < nbs) {
                    const int nb = 3 * h + q;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int n = nb * 32 + 8 * rr + 4 * g;
                        const float4 bb = *reinterpret_cast<const float4*>(Bs + n);
                        uint2 o;
                        o.x = pack_bf16x2(acc[nb][4 * rr + 0] + bb.x, acc[nb][4 * rr + 1] + bb.y);
                        o.y = pack_bf16x2(acc[nb][4 * rr + 2] + bb.z, acc[nb][4 * rr + 3] + bb.w);
                        *reinterpret_cast<uint2*>(strip + l31 * CD_STRIP_LD + q * 32 + 8 * rr + 4 * g) = o;
                        if (DUAL) {      // keep what was stored (bf16-rounded) for the norm
                            acc[nb][4 * rr + 0] = __uint_as_float(o.x << 16); acc[nb][4 * rr + 1] = __uint_as_float(o.x & 0xFFFF0000u);
                            acc[nb][4 * rr + 2] = __uint_as_float(o.y << 16); acc[nb][4 * rr + 3] = __uint_as_float(o.y & 0xFFFF0000u);
                        }
                    }
                }
            }
            // the strip is private to the wave: LDS operations of one wave complete in order, no barrier
            const int cpv = nbs * 4;                                               // 16-byte chunks per voxel in this pass
            auto store_strip = [&](u16* dst) {
                for (int j = lane; j < 32 * cpv; j += 64) {
                    const int v = j / cpv, ch = j - v * cpv;
                    const int64_t mv = vox0 + v;
                    if (mv < p.M) {
                        int tv = tf, rv = rem + v;
                        if (rv >= HWo) { rv -= HWo; ++tv; }
                        const int64_t vox = ((int64_t)(tv * p.ot_mul + p.ot_off)) * HWo + rv;
                        *reinterpret_cast<uint4*>(dst + vox * p.ldc + h * 96 + ch * 8) = *reinterpret_cast<const uint4*>(strip + v * CD_STRIP_LD + ch * 8);
                    }
                }
            };
            store_strip(p.y);
            if (DUAL) {
                // lane (l31, g) holds 48 of voxel l31's 96 channels, lane ^ 32 the others
                float ss = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) ss += acc[nb][e] * acc[nb][e];
                ss += __shfl_xor(ss, 32, 64);
                const float inv = sqrtf((float)p.N) * __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(ss), 1e-12f));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int n = nb * 32 + 8 * rr + 4 * g;
                        const float4 gm = *reinterpret_cast<const float4*>(Gs + n);
                        const float g4[4] = {gm.x, gm.y, gm.z, gm.w};
                        float v4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = acc[nb][4 * rr + e] * inv * g4[e];
                            v4[e] = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
                        }
                        uint2 o;
                        o.x = pack_bf16x2(v4[0], v4[1]);
                        o.y = pack_bf16x2(v4[2], v4[3]);
                        *reinterpret_cast<uint2*>(strip + l31 * CD_STRIP_LD + nb * 32 + 8 * rr + 4 * g) = o;
                    }
                store_strip(p.y2);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Halo-tile kernel for the 3x3x3 stride-1 causal convolutions (all ResidualBlock convs: 26 of the 33 decoder convs).
// (The 1x3x3 conv behind the 2x upsample uses it too, KT = 1 / UPS below; with ONE output frame per workgroup -- only 3 tap rows per patch
// load -- that was 8 % slower than the gather kernel, with two frames it is 65 % faster.)
// The gather kernel above fetches every input voxel once per tap (27x) and restages it through registers; here a
// workgroup owns an 8 x 16 block of ONE output frame, keeps the 3-frame (8+2) x (16+2) input patch of a CS-channel
// slice resident in LDS and reads the A fragments of every tap straight from it at the tap's offset: the patch is
// loaded once per slice (3.4x the block's own voxels instead of 27x) and never restaged.  Only the weights stream:
// per (dt, dh) "tap row" a [96 x (3 taps x CS k)] tile, register-staged.  4 waves, wave = 32 voxels (2 rows of the
// block) x 96 output channels.  Patch voxels and W rows are padded by 16 B: 16 consecutive voxels / rows start in 16
// distinct 16-byte bank groups.
//   CS = 48, two W buffers (one barrier per tap row, 27 MFMAs per wave between barriers): 118.8 KB LDS, 1 workgroup / CU
//   CS = 32, one W buffer (two barriers per tap row, 18 MFMAs):                             63.2 KB LDS, 2 workgroups / CU
//   CS = 32, one W buffer, NF = 2 output frames (36 MFMAs between barriers):                77.6 KB LDS, 2 workgroups / CU (default)
// Timing ablations of the NF = 1 kernel (96 channels, 512 x 896): no W global loads +16 %, no W loads / LDS stores +24 %,
// no patch reloads +7.5 %, W fragments read once instead of per 32-channel block +6 %, no barriers +3.4 % -- the W stream
// is what NF = 2 halves.
// ------------------------------------------------------------------------------------------------
#define HT_TH 8
#define HT_TW 16
#define HT_FVOX ((HT_TH + 2) * (HT_TW + 2))   // patch voxels per input frame
// SWZ: no padding; instead the 16-byte chunk index inside a voxel / W tap is XOR-ed with (index >> 2) & 3 of the voxel /
// row (needs CS == 32: 4 chunks): the two W buffers then fit next to the patch in half the LDS of a CU.
// NF: output frames per workgroup.  NF = 2 keeps a 4-frame patch and runs every W fragment against both frames: the W tile
// (the dominant L2 -> LDS stream: 0.5 MB per 128 output voxels at NF = 1) is staged and read once per 256 output voxels,
// the patch costs 4 input frames per 2 outputs instead of 3 per 1, and there are half as many barriers per MFMA.
// KT: temporal taps (3, or 1 for the 1x3x3 convolution behind the nearest 2x upsample: UPS, patch voxel (h, w) reads input
// (h >> 1, w >> 1); no temporal halo, 3 tap rows per slice).
template <int CS, int NWB, bool SWZ = false, int BN = 96, int NF = 1, int KT = 3, bool UPS = false, int PF = 0> struct HaloCfg {
    static constexpr int PS = SWZ ? CS : CS + 8, WS = SWZ ? 3 * CS : 3 * CS + 8;   // strides in elements
    static constexpr int PVOX = (NF + KT - 1) * HT_FVOX;
    static constexpr int PCH = PVOX * (CS / 8), WCH = BN * 3 * (CS / 8);
    static constexpr int NP = (PCH + 255) / 256, NWL = (WCH + 255) / 256;  // chunks per thread
    static constexpr int LDS = (PVOX * PS + NWB * BN * WS) * 2;
};

// PF: fragment prefetch.  0 = reads where the source puts them (hipcc issues them just in time: `ds_read; s_waitcnt; mfma`, the LDS
// latency of every group is exposed to the wave and only the SIMD's other wave covers it); 1 = the fragments of tap dw + 1 (2 k-steps x
// (NF + NBLK) reads) are requested, behind a scheduling barrier, before the MFMAs of tap dw run.
template <int EPI, int CS, int NWB, bool SWZ = false, int BN = 96, int NF = 1, int KT = 3, bool UPS = false, int PF = 0>   // BN = 96 (three 32-channel blocks) or 32 (narrow outputs: the RGB head)
__global__ __launch_bounds__(256, ((NWB == 1 || SWZ) ? 2 : 1)) void conv_halo_kernel(ConvParams p) {
    using Cfg = HaloCfg<CS, NWB, SWZ, BN, NF, KT>;
    constexpr int NBLK = BN / 32, HT_PVOX = Cfg::PVOX;
    static_assert(!SWZ || CS == 32, "the swizzle works on 4 chunks per voxel");
    constexpr int PS = Cfg::PS, WS = Cfg::WS, PCH = Cfg::PCH, WCH = Cfg::WCH, NP = Cfg::NP, NWL = Cfg::NWL, CPV = CS / 8;
    static_assert(NP <= 13 && NWL <= 7, "staging register sets below");
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* Ps = smem;                          // [3][TH+2][TW+2][PS]
    u16* Ws = smem + HT_PVOX * PS;           // [NWB][BN][WS]

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_w = (p.Wo + HT_TW - 1) / HT_TW, tiles_h = (p.Ho + HT_TH - 1) / HT_TH;
    int bid = blockIdx.x;
    const int tn = bid % tiles_n; bid /= tiles_n;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h;
    const int to = (bid / tiles_h) * NF;         // first output frame of this workgroup
    const int n0 = tn * BN, h0 = th * HT_TH, w0 = tw * HT_TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, g = lane >> 5;

    // ---- patch chunk c = tid + 256 i: voxel = c / CPV, 16-byte channel chunk = c % CPV ----
#define HT_PDECL(i_)                                                                                   \
    const int pc##i_ = tid + 256 * i_;                                                                 \
    int64_t psrc##i_ = -1; int pdst##i_ = 0;                                                           \
    if (i_ < NP && pc##i_ < PCH) {                                                                     \
        const int vox_ = pc##i_ / CPV, ch_ = pc##i_ - vox_ * CPV;                                      \
        const int dt_ = vox_ / HT_FVOX, rem_ = vox_ - dt_ * HT_FVOX;                                   \
        const int rr_ = rem_ / (HT_TW + 2), cc_ = rem_ - rr_ * (HT_TW + 2);                            \
        const int ti_ = to + dt_ - p.pt, hi_ = h0 + rr_ - p.ph, wi_ = w0 + cc_ - p.pw;                 \
        if (ti_ >= 0 && ti_ < p.Ti && hi_ >= 0 && hi_ < p.Ho && wi_ >= 0 && wi_ < p.Wo)                \
            psrc##i_ = (((int64_t)ti_ * p.Hi + (UPS ? hi_ >> 1 : hi_)) * p.Wi + (UPS ? wi_ >> 1 : wi_)) * p.Cin + ch_ * 8; \
        pdst##i_ = vox_ * PS + (SWZ ? ((ch_ ^ ((vox_ >> 2) & 3)) << 3) : ch_ * 8);                     \
    }                                                                                                  \
    uint4 pr##i_ = make_uint4(0, 0, 0, 0);
    HT_PDECL(0) HT_PDECL(1) HT_PDECL(2) HT_PDECL(3) HT_PDECL(4) HT_PDECL(5) HT_PDECL(6)
    HT_PDECL(7) HT_PDECL(8) HT_PDECL(9) HT_PDECL(10) HT_PDECL(11) HT_PDECL(12)
#define HT_PLOAD(i_, c0_) if (i_ < NP) pr##i_ = (psrc##i_ >= 0) ? *reinterpret_cast<const uint4*>(p.x + psrc##i_ + (c0_)) : make_uint4(0, 0, 0, 0);
#define HT_PSTORE(i_) if (i_ < NP && pc##i_ < PCH) *reinterpret_cast<uint4*>(Ps + pdst##i_) = pr##i_;
#define HT_P13(M_, ...) M_(0, ##__VA_ARGS__) M_(1, ##__VA_ARGS__) M_(2, ##__VA_ARGS__) M_(3, ##__VA_ARGS__) M_(4, ##__VA_ARGS__) \
    M_(5, ##__VA_ARGS__) M_(6, ##__VA_ARGS__) M_(7, ##__VA_ARGS__) M_(8, ##__VA_ARGS__) M_(9, ##__VA_ARGS__)                    \
    M_(10, ##__VA_ARGS__) M_(11, ##__VA_ARGS__) M_(12, ##__VA_ARGS__)

    // ---- W chunk c = tid + 256 i of a tap row: n = c / (3 CPV), dw = (c % (3 CPV)) / CPV, 16-byte chunk = c % CPV ----
#define HT_WDECL(i_)                                                                                   \
    const int wc##i_ = tid + 256 * i_;                                                                 \
    const int wn##i_ = wc##i_ / (3 * CPV), wrem##i_ = wc##i_ - wn##i_ * (3 * CPV), wdw##i_ = wrem##i_ / CPV, wch##i_ = wrem##i_ - wdw##i_ * CPV; \
    const u16* wsrc##i_ = p.w + (int64_t)min(n0 + wn##i_, p.N - 1) * p.Kpad + wdw##i_ * p.Cin + wch##i_ * 8; \
    const int wdst##i_ = wn##i_ * WS + wdw##i_ * CS + (SWZ ? ((wch##i_ ^ ((wn##i_ >> 2) & 3)) << 3) : wch##i_ * 8); \
    uint4 wr##i_ = make_uint4(0, 0, 0, 0);
    HT_WDECL(0) HT_WDECL(1) HT_WDECL(2) HT_WDECL(3) HT_WDECL(4) HT_WDECL(5) HT_WDECL(6)
#define HT_WLOAD(i_, k0_) if (i_ < NWL && wc##i_ < WCH) wr##i_ = *reinterpret_cast<const uint4*>(wsrc##i_ + (k0_));
#define HT_WSTORE(i_, buf_) if (i_ < NWL && wc##i_ < WCH) *reinterpret_cast<uint4*>(Ws + (buf_) * BN * WS + wdst##i_) = wr##i_;
#define HT_W7(M_, ...) M_(0, ##__VA_ARGS__) M_(1, ##__VA_ARGS__) M_(2, ##__VA_ARGS__) M_(3, ##__VA_ARGS__) M_(4, ##__VA_ARGS__) \
    M_(5, ##__VA_ARGS__) M_(6, ##__VA_ARGS__)

    // ---- fragment bases ----
    const int vloc = wave * 32 + l31;                                // voxel of the block: row vloc >> 4, col vloc & 15
    const int pv0 = (vloc >> 4) * (HT_TW + 2) + (vloc & 15);         // its patch voxel at tap (0, 0, 0)
    const u16* pa = Ps + pv0 * PS + (SWZ ? 0 : g * 8);
    const u16* wb = Ws + l31 * WS + (SWZ ? 0 : g * 8);
    int wsw[CS / 16];                                                // swizzled W chunk offset per k-step (row bits are lane constants)
#pragma unroll
    for (int ks = 0; ks < CS / 16; ++ks) wsw[ks] = SWZ ? (((ks * 2 + g) ^ ((l31 >> 2) & 3)) << 3) : ks * 16;

    f32x16 acc[NF][NBLK];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int a = 0; a < NBLK; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[f][a][e] = 0.f;

    const int nslice = p.Cin / CS;
    for (int sl = 0; sl < nslice; ++sl) {
        const int c0 = sl * CS;
        HT_P13(HT_PLOAD, c0)
        HT_W7(HT_WLOAD, c0)                          // tap row 0 of this slice
        __syncthreads();                             // every wave is done with the previous slice's patch and W tile
        HT_P13(HT_PSTORE)
        HT_W7(HT_WSTORE, 0)
        __syncthreads();
#pragma unroll 1
        for (int row = 0; row < 3 * KT; ++row) {     // row = dt * 3 + dh
            const int cur = (NWB == 2) ? (row & 1) : 0;
            if (row + 1 < 3 * KT) { HT_W7(HT_WLOAD, (row + 1) * 3 * p.Cin + c0) }
            const int dt = row / 3, dh = row - dt * 3;
            const int toff = (dt * (HT_TH + 2) + dh) * (HT_TW + 2);
            const u16* pr_ = pa + toff * PS;
            const u16* wr_ = wb + cur * BN * WS;
            if constexpr (PF == 0) {
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
#pragma unroll
                for (int ks = 0; ks < CS / 16; ++ks) {
                    bf16x8 xf[NF];
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const int psw = SWZ ? (((pv0 + toff + f * HT_FVOX + dw) >> 2) & 3) : 0;   // this tap's voxel: its swizzle bits
                        xf[f] = *reinterpret_cast<const bf16x8*>(pr_ + (f * HT_FVOX + dw) * PS + (SWZ ? (((ks * 2 + g) ^ psw) << 3) : ks * 16));
                    }
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb) {
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wr_ + nb * 32 * WS + dw * CS + wsw[ks]);
#pragma unroll
                        for (int f = 0; f < NF; ++f) acc[f][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[f], acc[f][nb], 0, 0, 0);
                    }
                }
            }
            } else {
                static_assert(!SWZ || PF == 0, "the prefetch variant uses the padded layout");
                constexpr int KS = CS / 16;
                bf16x8 xfb[2][KS][NF], wfb[2][KS][NBLK];
                auto fetch = [&](int b, int dw) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                        for (int f = 0; f < NF; ++f) xfb[b][ks][f] = *reinterpret_cast<const bf16x8*>(pr_ + (f * HT_FVOX + dw) * PS + ks * 16);
#pragma unroll
                        for (int nb = 0; nb < NBLK; ++nb) wfb[b][ks][nb] = *reinterpret_cast<const bf16x8*>(wr_ + nb * 32 * WS + dw * CS + wsw[ks]);
                    }
                };
                fetch(0, 0);
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    if (dw + 1 < 3) fetch((dw + 1) & 1, dw + 1);
                    __builtin_amdgcn_sched_barrier(0);       // the requests of tap dw + 1 stay in front of the MFMAs of tap dw
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                            for (int f = 0; f < NF; ++f)
                                acc[f][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfb[dw & 1][ks][nb], xfb[dw & 1][ks][f], acc[f][nb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (NWB == 1) __syncthreads();           // single buffer: every wave has read this row's tile
            if (row + 1 < 3 * KT) { HT_W7(HT_WSTORE, (NWB == 2) ? (cur ^ 1) : 0) }
            __syncthreads();
        }
    }

    // ---- epilogue: lane = voxel vloc, rows n = n0 + nb * 32 + 8 rr + 4 g + e ----
    const int ho = h0 + (vloc >> 4), wo = w0 + (vloc & 15);
    if (EPI == 4) {
        // conv -> RMS_norm -> SiLU in one pass (ResidualBlock residual.2 -> .3 -> .4, wan_vae.py:190-196): the workgroup's
        // single N tile holds every channel of a voxel, 48 (BN = 96) of them in this lane and the rest in lane ^ 32.
        // Same arithmetic as rms_silu_kernel on the bf16-rounded conv output: x * sqrt(C) / max(||x||, 1e-12) * gamma, SiLU.
        const float sC = sqrtf((float)p.N);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float ss = 0.f;
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int n = n0 + nb * 32 + 8 * rr + 4 * g;
                    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (n < p.N && p.bias != nullptr) bb = *reinterpret_cast<const float4*>(p.bias + n);
                    const float b4[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float vr = (n < p.N) ? bf2f(f2bf(acc[f][nb][4 * rr + e] + b4[e])) : 0.f;
                        acc[f][nb][4 * rr + e] = vr;
                        ss += vr * vr;
                    }
                }
            ss += __shfl_xor(ss, 32, 64);
            const float inv = sC / fmaxf(sqrtf(ss), 1e-12f);
            if (ho < p.Ho && wo < p.Wo && to + f < p.To) {
                const int64_t vox = ((int64_t)((to + f) * p.ot_mul + p.ot_off) * p.Ho + ho) * p.Wo + wo;
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int n = n0 + nb * 32 + 8 * rr + 4 * g;
                        if (n >= p.N) continue;
                        const float4 gm = *reinterpret_cast<const float4*>(p.gamma + n);
                        const float g4[4] = {gm.x, gm.y, gm.z, gm.w};
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_f(acc[f][nb][4 * rr + e] * inv * g4[e]);
                        uint2 o;
                        o.x = pack_bf16x2(v[0], v[1]);
                        o.y = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(p.y + vox * p.ldc + n) = o;
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f)
    if (ho < p.Ho && wo < p.Wo && to + f < p.To) {
        const int64_t vox = ((int64_t)((to + f) * p.ot_mul + p.ot_off) * p.Ho + ho) * p.Wo + wo;
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int n = n0 + nb * 32 + 8 * rr + 4 * g;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[f][nb][4 * rr + e];
                if (p.bias != nullptr) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (EPI == 3) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(p.resid + vox * p.ldr + n);
                    v[0] += bf_lo(rv.x); v[1] += bf_hi(rv.x); v[2] += bf_lo(rv.y); v[3] += bf_hi(rv.y);
                }
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(p.y + vox * p.ldc + n) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Halo-tile kernel for Resample's spatial downsampling (round 6): ZeroPad2d((0, 1, 0, 1)) + Conv2d(dim, dim, 3, stride 2) on every frame
// (wan_vae.py:87-96) = a 1 x 3 x 3 convolution with stride (1, 2, 2) whose padding sits behind the last row / column only.  The gather
// kernel ran the three of them (96 / 192 / 384 channels) at 360-600 TFLOP/s: one k-tile in flight per workgroup and every input voxel
// fetched once per tap that reads it.  Here a workgroup owns 8 x 16 output voxels of NF frames x 96 output channels and keeps the
// (2*8 + 1) x (2*16 + 1) input patch of a 32-channel slice in LDS, COLUMNS SPLIT BY PARITY (a patch row = its 17 even columns, then its 16
// odd ones): output column c reads input column 2 c + dw = parity dw & 1, index c + (dw >> 1), so the 16 lanes of an output row read 16
// CONSECUTIVE patch voxels for every tap (the padded 80-byte voxel stride keeps them in distinct bank groups, as in the stride-1 kernel; at a
// two-voxel stride they would collide in pairs).  W streams per tap row (3 taps x 32 k), register-staged, one row ahead; the NEXT slice's
// patch is requested right after the current one is in LDS and stays in flight under the slice's 54 x NF MFMAs per wave.  The n tiles of a
// voxel tile (192 / 384 channels) get workgroup ids on the SAME XCD, next to each other in time: its L2 serves the patch to all of them.
// ------------------------------------------------------------------------------------------------
#define S2_TH 8
#define S2_TW 16
#define S2_PH (2 * S2_TH + 1)
#define S2_PW (2 * S2_TW + 1)
#define S2_FVOX (S2_PH * S2_PW)
#define S2_CS 32
#define S2_PS (S2_CS + 8)
#define S2_WS (3 * S2_CS + 8)
template <int NF> struct S2Cfg {
    static constexpr int PVOX = NF * S2_FVOX, PCH = PVOX * (S2_CS / 8), NP = (PCH + 255) / 256;
    static constexpr int WCH = 96 * 3 * (S2_CS / 8), NWL = (WCH + 255) / 256;
    static constexpr int LDS = (PVOX * S2_PS + 96 * S2_WS) * 2 + 96 * 4;       // patch, W tap row, the tile's 96 bias values (fp32)
};

template <int NF>
__global__ __launch_bounds__(256, (NF == 1 ? 2 : 1)) void conv_s2_kernel(ConvParams p, int vtiles) {
    using Cfg = S2Cfg<NF>;
    constexpr int NP = Cfg::NP, NWL = Cfg::NWL, PCH = Cfg::PCH, WCH = Cfg::WCH, CPV = S2_CS / 8, NBLK = 3;
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* Ps = smem;                          // [NF][17][17 even | 16 odd columns][PS]
    u16* Ws = smem + Cfg::PVOX * S2_PS;      // [96][WS]
    float* Bs = reinterpret_cast<float*>(Ws + 96 * S2_WS);      // [96]: the epilogue's bias loads would each wait for L2 (hipcc serialises them)

    // workgroup id -> (voxel tile v, n tile): the XCD (id % 8) depends on v only
    const int tiles_n = p.N / 96;
    const int q = blockIdx.x >> 3;
    const int tn = q % tiles_n;
    int v = (q / tiles_n) * 8 + (blockIdx.x & 7);
    if (v >= vtiles) return;
    const int tiles_w = (p.Wo + S2_TW - 1) / S2_TW, tiles_h = (p.Ho + S2_TH - 1) / S2_TH;
    const int tw = v % tiles_w; v /= tiles_w;
    const int th = v % tiles_h;
    const int to = (v / tiles_h) * NF;       // first output (= input) frame of this workgroup
    const int n0 = tn * 96, h0 = th * S2_TH, w0 = tw * S2_TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, g = lane >> 5;

    // ---- patch chunk c = tid + 256 i (input order: frame, row, column, 16-byte channel chunk): element offset from frame `to`, LDS offset ----
    const u16* xb = p.x + (int64_t)to * p.Hi * p.Wi * p.Cin;
    int psrc[NP], pdst[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = tid + 256 * i;
        psrc[i] = -1; pdst[i] = 0;
        if (c < PCH) {
            const int vox = c / CPV, ch = c - vox * CPV;
            const int f = vox / S2_FVOX, rem = vox - f * S2_FVOX;
            const int rr = rem / S2_PW, cc = rem - rr * S2_PW;
            const int hi = 2 * h0 + rr, wi = 2 * w0 + cc;
            if (to + f < p.Ti && hi < p.Hi && wi < p.Wi) psrc[i] = ((f * p.Hi + hi) * p.Wi + wi) * p.Cin + ch * 8;
            pdst[i] = (f * S2_FVOX + rr * S2_PW + (cc & 1) * (S2_TW + 1) + (cc >> 1)) * S2_PS + ch * 8;
        }
    }
    // ---- W chunk c = tid + 256 i of a tap row: n = c / 12, dw = (c % 12) / 4, 16-byte chunk = c % 4 ----
    int wsrc[NWL], wdst[NWL];
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
        const int c = tid + 256 * i;
        const int n = c / (3 * CPV), rem = c - n * (3 * CPV), dw = rem / CPV, ch = rem - dw * CPV;
        wsrc[i] = (c < WCH) ? (n0 + n) * p.Kpad + dw * p.Cin + ch * 8 : 0;       // (past the tile: loads a valid address, never stored)
        wdst[i] = n * S2_WS + dw * S2_CS + ch * 8;
    }
    const bool wtail = tid + 256 * (NWL - 1) < WCH;                               // the last chunk exists for this thread
    uint4 pr[NP], wr[3][NWL];                                                      // W: one register set per tap row of a slice, requested two rows ahead
#pragma unroll
    for (int i = 0; i < NP; ++i) pr[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < NWL; ++i) wr[r][i] = make_uint4(0, 0, 0, 0);
    auto load_patch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) pr[i] = *(psrc[i] >= 0 ? reinterpret_cast<const uint4*>(xb + psrc[i] + c0) : &g_conv_zero16);     // (no branch: padding reads a zero block)
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (tid + 256 * i < PCH) *reinterpret_cast<uint4*>(Ps + pdst[i]) = pr[i];
    };
    auto load_w = [&](int r, int k0) {
#pragma unroll
        for (int i = 0; i < NWL; ++i) wr[r][i] = *reinterpret_cast<const uint4*>(p.w + wsrc[i] + k0);
    };
    auto store_w = [&](int r) {
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            if (i + 1 < NWL || wtail) *reinterpret_cast<uint4*>(Ws + wdst[i]) = wr[r][i];
    };

    // ---- fragment bases: lane = output voxel vloc (row vloc >> 4, column vloc & 15) of the block ----
    const int vloc = wave * 32 + l31;
    const u16* pa = Ps + (2 * (vloc >> 4) * S2_PW + (vloc & 15)) * S2_PS + g * 8;       // its patch voxel at tap (0, 0)
    const u16* wb = Ws + l31 * S2_WS + g * 8;

    f32x16 acc[NF][NBLK];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int a = 0; a < NBLK; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[f][a][e] = 0.f;

    const int nslice = p.Cin / S2_CS;
    load_patch(0);
    load_w(0, 0);
    load_w(1, 3 * p.Cin);
    if (tid < 96) Bs[tid] = p.bias != nullptr ? p.bias[n0 + tid] : 0.f;
    for (int sl = 0; sl < nslice; ++sl) {
        const int c0 = sl * S2_CS;
        __syncthreads();                             // every wave is done with the previous slice's patch and W tile
        store_patch();
        store_w(0);
        __syncthreads();
        if (sl + 1 < nslice) load_patch(c0 + S2_CS);   // in flight under this slice's MFMAs
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            // tap row dh + 2 of the slice-major row sequence: row 2 of this slice, rows 0 / 1 of the next
            if (dh == 0) load_w(2, 6 * p.Cin + c0);
            else if (sl + 1 < nslice) load_w(dh - 1, (dh - 1) * 3 * p.Cin + c0 + S2_CS);
            __builtin_amdgcn_sched_barrier(0);       // (hipcc otherwise sinks the requests behind the tap row's MFMAs, right in front of the barrier that waits for them)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int toff = dh * S2_PW + (dw & 1) * (S2_TW + 1) + (dw >> 1);
#pragma unroll
                for (int ks = 0; ks < S2_CS / 16; ++ks) {
                    bf16x8 xf[NF];
#pragma unroll
                    for (int f = 0; f < NF; ++f) xf[f] = *reinterpret_cast<const bf16x8*>(pa + (f * S2_FVOX + toff) * S2_PS + ks * 16);
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb) {
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + nb * 32 * S2_WS + dw * S2_CS + ks * 16);
#pragma unroll
                        for (int f = 0; f < NF; ++f) acc[f][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[f], acc[f][nb], 0, 0, 0);
                    }
                }
            }
            if (dh < 2) {
                __syncthreads();                     // every wave has read this tap row's W tile
                store_w(dh + 1);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: lane = voxel vloc, rows n = n0 + nb * 32 + 8 rr + 4 g + e ----
    const int ho = h0 + (vloc >> 4), wo = w0 + (vloc & 15);
#pragma unroll
    for (int f = 0; f < NF; ++f)
        if (ho < p.Ho && wo < p.Wo && to + f < p.To) {
            const int64_t vox = ((int64_t)((to + f) * p.ot_mul + p.ot_off) * p.Ho + ho) * p.Wo + wo;
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int n = n0 + nb * 32 + 8 * rr + 4 * g;
                    const float4 bb = *reinterpret_cast<const float4*>(Bs + nb * 32 + 8 * rr + 4 * g);
                    uint2 o;
                    o.x = pack_bf16x2(acc[f][nb][4 * rr + 0] + bb.x, acc[f][nb][4 * rr + 1] + bb.y);
                    o.y = pack_bf16x2(acc[f][nb][4 * rr + 2] + bb.z, acc[f][nb][4 * rr + 3] + bb.w);
                    *reinterpret_cast<uint2*>(p.y + vox * p.ldc + n) = o;
                }
        }
}

// ------------------------------------------------------------------------------------------------
// RMS_norm over channels (F.normalize * sqrt(C) * gamma, wan_vae.py:39-54) + optional SiLU.
// LPV lanes per voxel (power of two >= C/8), 64/LPV voxels per wave.
// ------------------------------------------------------------------------------------------------
// LPV lanes per voxel (power of two), CPL 16-byte chunks per lane (lane j owns chunks j, j + LPV, ...: C = 96 -> 4 lanes
// x 3 chunks, all lanes busy), VPT voxels per lane group with every load issued before the first use.
template <int CPL, int VPT>
__global__ __launch_bounds__(256) void rms_silu_kernel(const u16* __restrict__ x, u16* __restrict__ y,
                                                       const float* __restrict__ gamma, int64_t nvox, int C,
                                                       int lpv_log2, int silu) {
    // persistent blocks (round 5): a lane keeps its gamma chunks in registers and walks over voxel groups -- the one-shot form issued 8 scalar
    // gamma loads per 16-byte data load and ran at 2.5 TB/s (read + write) where the row passes of the DiT reach 5.2
    const int lpv = 1 << lpv_log2;
    const int sub = (int)(threadIdx.x & (lpv - 1));
    const int nch = C >> 3;
    float gm[CPL][8];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = min(sub + c * lpv, nch - 1);
        if ((reinterpret_cast<uintptr_t>(gamma) & 15) == 0) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch * 8), g1 = *reinterpret_cast<const float4*>(gamma + ch * 8 + 4);
            gm[c][0] = g0.x; gm[c][1] = g0.y; gm[c][2] = g0.z; gm[c][3] = g0.w;
            gm[c][4] = g1.x; gm[c][5] = g1.y; gm[c][6] = g1.z; gm[c][7] = g1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) gm[c][e] = gamma[ch * 8 + e];
        }
    }
    const float sC = sqrtf((float)C);
    const int64_t gpb = blockDim.x >> lpv_log2;                 // voxel groups (of VPT voxels) per block and iteration
    for (int64_t grp = (int64_t)blockIdx.x * gpb + (threadIdx.x >> lpv_log2); grp * VPT < nvox; grp += (int64_t)gridDim.x * gpb) {
    const int64_t vox0 = grp * VPT;
    uint4 raw[VPT][CPL];
#pragma unroll
    for (int j = 0; j < VPT; ++j)
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = sub + c * lpv;
            raw[j][c] = (vox0 + j < nvox && ch < nch) ? *reinterpret_cast<const uint4*>(x + (vox0 + j) * C + ch * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        float v[CPL][8];
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            unpack8(raw[j][c], v[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) q += v[c][e] * v[c][e];
        }
        for (int o = 1; o < lpv; o <<= 1) q += __shfl_xor(q, o, 64);
        // v_sqrt / v_rcp / v_exp forms (1 ulp), the arithmetic of the generated kernels' fused norm epilogue (asmgen/conv4.py norm_silu): the
        // IEEE division sequences of `sC / x` and `t / (1 + exp(-t))` were 2/3 of this kernel's VALU work (19 -> 8 instructions per element)
        const float inv = sC * __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(q), 1e-12f));
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = sub + c * lpv;
            if (vox0 + j < nvox && ch < nch) {
                float o8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = v[c][e] * inv * gm[c][e];
                    o8[e] = silu ? t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t)) : t;
                }
                *reinterpret_cast<uint4*>(y + (vox0 + j) * C + ch * 8) = pack8(o8);
            }
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// softmax over rows in place (mid-block attention, wan_vae.py:252), scale folded in; n <= 8192
// ------------------------------------------------------------------------------------------------
// n valid columns (any n >= 1; the row is padded to a multiple of 8 in memory): columns n .. ceil8(n) are written as zeros.
// MAXV 16-byte chunks per thread: 4 -> n <= 8192 (the 512p mid block: 7168 tokens per frame), 16 -> n <= 32768.
template <int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(u16* __restrict__ s, int64_t ld, int n, float scale) {
    __shared__ float red[4];
    u16* row = s + (int64_t)blockIdx.x * ld;
    const int nvec = (n + 7) >> 3;
    float v[MAXV][8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < nvec) {
            unpack8(*reinterpret_cast<const uint4*>(row + c * 8), v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (c * 8 + e >= n) v[j][e] = -INFINITY;      // ragged tail
                mx = fmaxf(mx, v[j][e]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sl2 = scale * 1.4426950408889634f;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[j][e] = __builtin_amdgcn_exp2f((v[j][e] - mx) * sl2);      // exp2(-inf) = 0 for the tail
                sum += v[j][e];
            }
        }
    }
    sum = block_sum_256(sum, red);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] *= inv;
            *reinterpret_cast<uint4*>(row + c * 8) = pack8(v[j]);
        }
    }
}

// batched 2-D transpose  (R x C, row stride ldi) -> (C x R), 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void transpose2d_kernel(const u16* __restrict__ in, int64_t ldi, int64_t in_bs,
                                                          u16* __restrict__ out, int64_t ldo, int64_t out_bs, int R, int C) {
    __shared__ u16 tile[64][66];
    const u16* src = in + (int64_t)blockIdx.z * in_bs;
    u16* dst = out + (int64_t)blockIdx.z * out_bs;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < R && c0 + c < C) ? src[(int64_t)(r0 + r) * ldi + c0 + c] : (u16)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < C && r0 + r < R) dst[(int64_t)(c0 + c) * ldo + r0 + r] = tile[r][c];
    }
}

// planar fp32 (C, N) -> channels-last bf16 (N, Cpad) with per-channel affine; channels >= C are zero
__global__ void to_channels_last_kernel(const float* __restrict__ x, u16* __restrict__ y, const float* __restrict__ a,
                                        const float* __restrict__ b, int C, int Cpad, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Cpad) return;
    const int c = (int)(i % Cpad);
    const int64_t n = i / Cpad;
    float v = 0.f;
    if (c < C) v = x[(int64_t)c * N + n] * (a ? a[c] : 1.f) + (b ? b[c] : 0.f);
    y[i] = f2bf(v);
}

// channels-last bf16 (N, ldx) -> planar fp32 (C, N): y = clamp((x + b) * a, lo, hi)
__global__ void from_channels_last_kernel(const u16* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                          const float* __restrict__ a, const float* __restrict__ b, int C, int64_t N,
                                          float lo, float hi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int64_t n = i % N;
    const int c = (int)(i / N);
    float v = (bf2f(x[n * ldx + c]) + (b ? b[c] : 0.f)) * (a ? a[c] : 1.f);
    y[i] = fminf(fmaxf(v, lo), hi);
}

// ================================================================================================
// C ABI
// ================================================================================================
// ------------------------------------------------------------------------------------------------
// conv4: the hand-scheduled 3x3x3 kernels (2 frames x 16 x 16 voxels x 96 channels per workgroup, one wave per SIMD, chunk-planar
// patch in LDS written by LDS-DMA with the padding supplied by the buffer descriptors' range check).  gfx950 assembly GENERATED by
// scail_amd/asmgen/conv4.py (csrc/conv4.s), embedded as a code object.  Kernel argument block = asmgen/conv4.py KERNARG_FMT.
// ------------------------------------------------------------------------------------------------
static const unsigned char k_conv4_hsaco[] = {
#include "conv4_hsaco.inc"
};
// the kt = 1 kernels (scail_conv4u_*: the 1x3x3 convolution of Resample, optionally behind the nearest 2x upsample; Cfg.kt = 1) and the narrow-output
// 3x3x3 kernel (scail_conv4n_*: N <= 16, the RGB head; Cfg.nb = 1) and conv + RMS_norm + SiLU in one kernel (scail_conv4f_e4: N = 96): csrc/conv4u.s
static const unsigned char k_conv4u_hsaco[] = {
#include "conv4u_hsaco.inc"
};
struct Conv4Args {
    const void* x; const void* w; const void* bias; void* y; const void* resid;
    int32_t Ti, To, H, W;
    int32_t Cin, N, Kpad, pt;
    int32_t tiles_t, tiles_w, tiles_n; uint32_t magic_n;      // frame pairs, tile columns, n tiles (digits of the tile number: n tile fastest, then frame pair, column, row)
    uint32_t magic_w, magic_t; int32_t n_slices, ot_mul;
    int32_t ot_off, tiles_per_wg;
    int64_t ldc, ldr;
    int32_t wgs_per_xcd, tiles;      // persistent workgroups per XCD (the tile stride of a workgroup), tiles in all
    const void* gamma;               // (scail_conv4c_e5 / e6) RMS_norm weights of the consumer behind the residual sum, fp32 [96]
    int64_t y2_delta;                // (scail_conv4c_e5) bytes from y (the raw sum) to the normalised copy: same row stride and frame mapping
};
static_assert(sizeof(Conv4Args) == 152, "Conv4Args must match asmgen/conv4.py KERNARG_SIZE");
static std::map<std::pair<int, int>, hipModule_t> g_conv4_modules;          // (device, code object: 0 conv4.s, 1 conv4u.s) -> loaded module
static std::map<std::pair<int, std::string>, hipFunction_t> g_conv4_fn;     // (device, kernel name)
static std::mutex g_conv4_mutex;
static std::string g_conv4_suffix;                                          // measurement build: "conv4_kernel:<suffix>"

static int conv4_function(const std::string& name, hipFunction_t* fn) {
    std::lock_guard<std::mutex> lk(g_conv4_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        scail_set_error("conv4: hipGetDevice failed");
        return 2;
    }
    const int which = (name.rfind("scail_conv4u", 0) == 0 || name.rfind("scail_conv4n", 0) == 0 || name.rfind("scail_conv4f", 0) == 0 ||
                       name.rfind("scail_conv4c", 0) == 0) ? 1 : 0;
    auto mit = g_conv4_modules.find(std::make_pair(dev, which));
    if (mit == g_conv4_modules.end()) {
        hipModule_t mod = nullptr;
        hipError_t e = hipModuleLoadData(&mod, which ? k_conv4u_hsaco : k_conv4_hsaco);
        if (e != hipSuccess) {
            scail_set_error(std::string("conv4: hipModuleLoadData failed: ") + hipGetErrorString(e));
            return 2;
        }
        mit = g_conv4_modules.emplace(std::make_pair(dev, which), mod).first;
    }
    auto it = g_conv4_fn.find(std::make_pair(dev, name));
    if (it == g_conv4_fn.end()) {
        hipFunction_t f;
        hipError_t e = hipModuleGetFunction(&f, mit->second, name.c_str());
        if (e != hipSuccess) {
            scail_set_error("conv4: kernel " + name + " is not in the embedded code object: " + hipGetErrorString(e));
            return 2;
        }
        it = g_conv4_fn.emplace(std::make_pair(dev, name), f).first;
    }
    *fn = it->second;
    return 0;
}

// compute units of the current device, cached per HIP device id (like gemm4_cu_count)
static int conv4_cu_count() {
    static std::map<int, int> cache;
    static std::mutex mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(dev);
    if (it == cache.end()) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        it = cache.emplace(dev, n).first;
    }
    return it->second;
}

static std::atomic<int> g_conv4_cont{1};                                   // option "conv4_cont": the tile-continuation variants (scail_conv4c_e0 / e3 / e4) for the one-n-tile shapes
int scail_conv4_cont_enable(int v) { g_conv4_cont = v != 0; return 0; }
static std::atomic<int> g_conv4_resnorm{1};                                // option "conv4_resnorm": scail_conv3d_cl_resid_norm on the generated dual-output / norm-only epilogues (scail_conv4c_e5 / e6)
int scail_conv4_resnorm_enable(int v) { g_conv4_resnorm = v != 0; return 0; }
static std::atomic<int> g_conv_direct{1};                                  // option "conv_direct": the direct-gather kernel for the HBM-bound convolutions
int scail_conv_direct_enable(int v) { g_conv_direct = v != 0; return 0; }
static std::atomic<int> g_conv4{1};                                        // option "conv4": the generated kernels where scail_conv3d_kernel_for says 4
static std::atomic<int> g_conv_s2{1};                                      // option "conv_s2": the stride-2 halo kernel for Resample's downsampling convolution (n = output frames per workgroup; 0: gather kernel)
int scail_conv_s2_enable(int v) { g_conv_s2 = v < 0 ? 0 : (v > 2 ? 2 : v); return 0; }
int scail_conv4_enable(int v) { g_conv4 = v != 0; return 0; }
#ifdef SCAIL_ABLATIONS
static int g_conv_halo = 4;                                    // measurement build: A/B of the halo-kernel layouts (comment below)
int scail_conv_tune(int v) { if (v >= 10) g_conv4 = v - 10; else g_conv_halo = v; return 0; }     // 10 / 11: generated kernel off / on
int scail_conv4_kernel(const char* suffix) { g_conv4_suffix = suffix ? suffix : ""; if (!g_conv4_suffix.empty() && g_conv4_suffix[0] == ':') g_conv4_suffix = "_" + g_conv4_suffix.substr(1); return 0; }
#else
static constexpr int g_conv_halo = 4;
#endif

// the shapes the generated kernels cover: 3x3x3, stride 1, 'same' spatial extent, 0..2 padding frames in front, whole 32-channel slices and
// 96-channel output tiles, at least one frame pair, 32-bit byte offsets inside a frame, and a tile grid whose id decode is exact: the kernel
// divides a tile id by tiles_n, tiles_t and tiles_w with 31-bit magic numbers, exact while dividend x divisor < 2^31
static int64_t conv4_tiles(const ConvParams& p) { return (int64_t)((p.To + 1) / 2) * ((p.Ho + 15) / 16) * ((p.Wo + 15) / 16) * std::max(p.N / 96, 1); }
static bool conv4_eligible(const ConvParams& p, int64_t ldc, int64_t ldr) {
    if (p.N <= 0 || p.N % 96 != 0 || p.To < 2 || p.Ho <= 0 || p.Wo <= 0) return false;
    const int64_t max_div = std::max<int64_t>({p.N / 96, (p.To + 1) / 2, (p.Wo + 15) / 16});
    if (conv4_tiles(p) * max_div >= (1ll << 31)) return false;
    return p.kt == 3 && p.kh == 3 && p.kw == 3 && p.st == 1 && p.sh == 1 && p.sw == 1 && !p.ups && p.ph == 1 && p.pw == 1 &&
           p.Ho == p.Hi && p.Wo == p.Wi && p.Cin % 32 == 0 && p.N % 96 == 0 && p.To >= 2 && p.pt >= 0 && p.pt <= 2 &&
           ldc % 8 == 0 && ldr % 8 == 0 && ldc < (1 << 20) && ldr < (1 << 20) && (int64_t)p.Hi * p.Wi * p.Cin * 2 < (1ll << 31) &&
           (int64_t)p.Ho * p.Wo * std::max(ldc, ldr) * 2 < (1ll << 32);
}

// the kt = 1 kernels: 1 x 3 x 3, stride 1, padding (0, 1, 1), every output frame from the input frame of the same index, same extent or
// (ups) exactly twice the input's; otherwise the limits above (32-bit byte offsets inside an INPUT frame)
static bool conv4u_eligible(const ConvParams& p, int64_t ldc) {
    if (p.N <= 0 || p.N % 96 != 0 || p.To < 2 || p.Ho <= 0 || p.Wo <= 0) return false;
    const int64_t max_div = std::max<int64_t>({p.N / 96, (p.To + 1) / 2, (p.Wo + 15) / 16});
    if (conv4_tiles(p) * max_div >= (1ll << 31)) return false;
    const bool extent = p.ups ? (p.Ho == 2 * p.Hi && p.Wo == 2 * p.Wi) : (p.Ho == p.Hi && p.Wo == p.Wi);
    return p.kt == 1 && p.kh == 3 && p.kw == 3 && p.st == 1 && p.sh == 1 && p.sw == 1 && p.pt == 0 && p.ph == 1 && p.pw == 1 && extent &&
           p.Ti == p.To && p.Cin % 32 == 0 && ldc % 8 == 0 && ldc < (1 << 20) && (int64_t)p.Hi * p.Wi * p.Cin * 2 < (1ll << 31) &&
           (int64_t)p.Ho * p.Wo * ldc * 2 < (1ll << 32);
}

// the narrow-output kernel: the 3x3x3 shapes of conv4_eligible with N = 8 or 16 output channels (one n tile, W rows past N read zeros), no
// residual; 8-byte stores (ldc % 4 == 0)
static bool conv4n_eligible(const ConvParams& p, int64_t ldc) {
    if (p.N != 8 && p.N != 16) return false;
    if (p.To < 2 || p.Ho <= 0 || p.Wo <= 0) return false;
    const int64_t max_div = std::max<int64_t>((p.To + 1) / 2, (p.Wo + 15) / 16);
    if (conv4_tiles(p) * max_div >= (1ll << 31)) return false;
    return p.kt == 3 && p.kh == 3 && p.kw == 3 && p.st == 1 && p.sh == 1 && p.sw == 1 && !p.ups && p.ph == 1 && p.pw == 1 &&
           p.Ho == p.Hi && p.Wo == p.Wi && p.Cin % 32 == 0 && p.pt >= 0 && p.pt <= 2 && ldc % 4 == 0 && ldc >= p.N && ldc < (1 << 20) &&
           (int64_t)p.Hi * p.Wi * p.Cin * 2 < (1ll << 31) && (int64_t)p.Ho * p.Wo * ldc * 2 < (1ll << 32);
}

// the direct-gather kernel's dual-output form (conv_direct_kernel<14, 3, true>): 96 output channels, up to 14 k-steps of 16 (the encoder's stem: 3 x 3 x 3 taps
// of 8 padded channels), no residual, no upsample; shape conditions of the plain dispatch in conv3d_impl
static bool conv_direct_dual_eligible(const ConvParams& p, int64_t ldc) {
    const int ksteps = (p.Ktrue + 15) / 16;
    const int wld = ksteps * 16 + 8;
    const int lds = (p.N * wld + 8 * 32 * CD_STRIP_LD) * 2 + 2 * p.N * 4;
    return p.N == 96 && ksteps <= 14 && !p.ups && p.Cin % 8 == 0 && p.kt <= 30 && p.kh <= 31 && p.kw <= 31 && p.Cin < 65536 &&
           (int64_t)(p.kt + 1) * p.Hi * p.Wi * p.Cin < (1ll << 31) && lds <= 150 * 1024 && ldc % 8 == 0 && p.M >= 4096 && p.M < (1ll << 40) &&
           p.Ho * (int64_t)p.Wo < (1ll << 31) && p.Ho * (int64_t)p.Wo >= 32;
}

static void conv_params(ConvParams& p, const int32_t* geom) {
    p.Ti = geom[0]; p.Hi = geom[1]; p.Wi = geom[2]; p.Cin = geom[3];
    p.To = geom[4]; p.Ho = geom[5]; p.Wo = geom[6];
    p.kt = geom[7]; p.kh = geom[8]; p.kw = geom[9];
    p.st = geom[10]; p.sh = geom[11]; p.sw = geom[12];
    p.pt = geom[13]; p.ph = geom[14]; p.pw = geom[15];
    p.ups = geom[16]; p.ot_mul = geom[17]; p.ot_off = geom[18];
    p.N = geom[19]; p.Kpad = geom[20];
    p.Ktrue = p.kt * p.kh * p.kw * p.Cin;
    p.M = (int64_t)p.To * p.Ho * p.Wo;
}

extern "C" int scail_conv3d_kernel_for(const int32_t* geom, int64_t ldc, int64_t ldr, int fused_norm) {
    if (geom == nullptr) return 0;
    ConvParams p;
    conv_params(p, geom);
    if (fused_norm == 2) { // (residual sum +) the NEXT consumer's RMS_norm + SiLU (scail_conv3d_cl_resid_norm): one n tile of 96 channels; with a residual the
                           // 3x3x3 kernels with tile continuation (scail_conv4c_e5 / e6), without one the kt = 1 kernel (scail_conv4u_e7)
        if (!(g_conv4 && g_conv4_resnorm && g_conv4_suffix.empty() && p.N == 96 && ldc == 96 && p.ot_mul == 1 && p.ot_off == 0)) return 0;
        if (ldr > 0) return (g_conv4_cont && conv4_eligible(p, ldc, ldr)) ? 4 : 0;
        if (conv4u_eligible(p, ldc)) return 4;
        // ... or, answer 2, the hipcc direct-gather kernel's dual-output form (the encoder's stem convolution): not a generated kernel, one launch all the same
        return (g_conv_direct && !conv4_eligible(p, ldc, ldc) && !conv4n_eligible(p, ldc) && conv_direct_dual_eligible(p, ldc)) ? 2 : 0;
    }
    if (fused_norm)      // conv + RMS_norm + SiLU: the generated kernel where one n tile holds a voxel's 96 channels, no residual
        return (g_conv4 && ldr == 0 && p.N == 96 && conv4_eligible(p, ldc, ldc)) ? 4 : 0;
    return (g_conv4 && (conv4_eligible(p, ldc, ldr > 0 ? ldr : ldc) || (ldr == 0 && (conv4u_eligible(p, ldc) || conv4n_eligible(p, ldc))))) ? 4 : 0;
}

// next_gamma / y_norm (scail_conv3d_cl_resid_norm, generated kernels only -- the caller checked scail_conv3d_kernel_for(.., 2)): the residual sum
// goes to y (nullptr: nowhere) and SiLU(RMS_norm(bf16(sum)) * next_gamma) to y_norm
static int conv3d_impl(const scail_bf16* x, const scail_bf16* w, const float* bias, scail_bf16* y, int64_t ldc,
                       const scail_bf16* resid, int64_t ldr, const float* gamma, const int32_t* geom, void* stream,
                       const float* next_gamma = nullptr, scail_bf16* y_norm = nullptr) {
    // geom: Ti Hi Wi Cin | To Ho Wo | kt kh kw | st sh sw | pt ph pw | ups | ot_mul ot_off | N Kpad
    ConvParams p;
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.ldc = ldc; p.resid = resid; p.ldr = ldr; p.gamma = gamma;
    conv_params(p, geom);
    SCAIL_REQUIRE(p.Cin % 8 == 0, "Cin must be a multiple of 8 (pad the channels)");
    SCAIL_REQUIRE(p.N % 8 == 0 && p.Kpad % CBK == 0 && p.Kpad >= p.Ktrue, "N % 8 == 0, Kpad % 64 == 0, Kpad >= taps*Cin");
    SCAIL_REQUIRE(ldc % 4 == 0 && (resid == nullptr || ldr % 4 == 0), "output / residual row strides must be multiples of 4");
    const bool rn = next_gamma != nullptr;
    if (rn && y == nullptr) { y = y_norm; p.y = y_norm; }      // norm-only form: the one output IS the normalised tensor
    SCAIL_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(y) & 7) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0,
                  "pointer alignment");
    if (p.M == 0) return 0;
    // 3x3x3 stride-1 causal convolutions: halo-tile kernel.  Knob conv_halo: 0 off, 1 = 48-channel slices / 1 workgroup
    // per CU, 4 (default) = 3 with TWO output frames per workgroup (4-frame patch, every W fragment used for both frames:
    // +22-25 % on the 96 / 192 / 384-channel shapes, bit-identical), 3 = 32-channel slices, one padded W buffer (two
    // barriers per tap row), 2 workgroups per CU,
    // 2 = 32-channel slices, swizzled unpadded layout with two W buffers (one barrier per tap row), 2 workgroups per CU --
    // measured equal (283 / 490 ms vs 281 / 487 ms encode / decode): the barrier is not what bounds the kernel
    const bool fuse = gamma != nullptr;       // conv + RMS_norm + SiLU: always the halo kernel, whatever the knob says
    const bool k1 = g_conv4 && !fuse && resid == nullptr && conv4u_eligible(p, ldc) && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    const bool nar = g_conv4 && !fuse && resid == nullptr && conv4n_eligible(p, ldc) && (reinterpret_cast<uintptr_t>(y) & 15) == 0;   // (8-byte stores)
    // conv -> RMS_norm -> SiLU (scail_conv3d_cl_norm) on the generated kernel: one n tile of 96 channels; gamma travels in the residual argument
    const bool fnorm = g_conv4 && fuse && resid == nullptr && p.N == 96 && conv4_eligible(p, ldc, ldc) && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(gamma) & 15) == 0;
    if (k1 || nar || fnorm || (g_conv4 && !fuse && conv4_eligible(p, ldc, resid ? ldr : ldc) && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(resid) & 15) == 0)) {      // (16-byte row chunks; the arena's tensors always are)
        Conv4Args a;
        a.x = x; a.w = w; a.bias = bias; a.y = y; a.resid = fnorm ? static_cast<const void*>(gamma) : static_cast<const void*>(resid);
        // (kt = 1 kernels: H, W are the OUTPUT extent, the `pt` argument carries the upsample shift -- there are no padding frames)
        a.Ti = p.Ti; a.To = p.To; a.H = p.Ho; a.W = p.Wo; a.Cin = p.Cin; a.N = p.N; a.Kpad = p.Kpad; a.pt = k1 ? (p.ups ? 1 : 0) : p.pt;
        a.tiles_t = (p.To + 1) / 2; a.tiles_w = (p.Wo + 15) / 16; a.tiles_n = std::max(p.N / 96, 1);
        auto magic31 = [](int d) { return (uint32_t)(((1ull << 31) + (uint64_t)d - 1) / (uint64_t)d); };
        a.magic_n = magic31(a.tiles_n); a.magic_w = magic31(a.tiles_w); a.magic_t = magic31(a.tiles_t);
        a.n_slices = p.Cin / 32; a.ot_mul = p.ot_mul; a.ot_off = p.ot_off; a.ldc = ldc; a.ldr = resid ? ldr : ldc;
        const int64_t tiles = conv4_tiles(p);        // conv4_eligible: tiles x the largest divisor < 2^31, so the magic-number divisions are exact
        // persistent workgroups, one per compute unit; tiles are numbered n tile fastest, then frame pair; workgroup number w = (b % 8) *
        // wgs_per_xcd + b / 8.  One n tile: w takes the next tiles_per_wg (+ 1) tiles (the frame pairs of a spatial tile; consecutive pairs
        // share two input frames, the lane offsets stay).  Several n tiles (tiles_per_wg = 0): w takes tiles w, w + grid, ... so that an XCD
        // works on the n tiles of a few neighbouring frame pairs at a time and they share the patch through its L2.
        const int cus = conv4_cu_count();
        a.wgs_per_xcd = (int32_t)std::min<int64_t>((tiles + 7) / 8, cus / 8);
        a.tiles_per_wg = a.tiles_n == 1 ? (int32_t)(tiles / (8 * a.wgs_per_xcd)) : 0;
        a.tiles = (int32_t)tiles;
        a.gamma = next_gamma;
        a.y2_delta = (rn && y != y_norm) ? (int64_t)(reinterpret_cast<const char*>(y_norm) - reinterpret_cast<const char*>(y)) : 0;
        hipFunction_t fn;
        // (measurement build: the "_prof" variant is an e0 kernel that writes its phase timers through the residual pointer)
        const bool prof = g_conv4_suffix.find("prof") != std::string::npos;
        const bool cont = g_conv4_cont && !k1 && !nar && a.tiles_n == 1 && g_conv4_suffix.empty();      // one n tile: runs of frame pairs per workgroup (a measurement-build kernel variant takes precedence)
        if (int rc = rn ? conv4_function(k1 ? "scail_conv4u_e7" : y != y_norm ? "scail_conv4c_e5" : "scail_conv4c_e6", &fn)
                        : k1 ? conv4_function("scail_conv4u_e0", &fn) : nar ? conv4_function(g_conv4_cont && g_conv4_suffix.empty() ? "scail_conv4cn_e0" : "scail_conv4n_e0", &fn)
                        : fnorm ? conv4_function(cont ? "scail_conv4c_e4" : "scail_conv4f_e4", &fn)
                        : cont ? conv4_function(resid ? "scail_conv4c_e3" : "scail_conv4c_e0", &fn)
                        : conv4_function(std::string(resid && !prof ? "scail_conv4_e3" : "scail_conv4_e0") + g_conv4_suffix, &fn)) return rc;
        size_t sz = sizeof(a);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        hipError_t e = hipModuleLaunchKernel(fn, (unsigned)a.wgs_per_xcd * 8u, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, extra);
        if (e != hipSuccess) {
            scail_set_error(std::string("conv4: launch failed: ") + hipGetErrorString(e));
            return 2;
        }
        return 0;
    }
    if (!rn && (g_conv_halo || fuse) && p.kt == 3 && p.kh == 3 && p.kw == 3 && p.st == 1 && p.sh == 1 && p.sw == 1 && !p.ups &&
        p.ph == 1 && p.pw == 1 && p.Ho == p.Hi && p.Wo == p.Wi &&
        p.Cin % ((g_conv_halo == 1 && p.N > 32 && !fuse) ? 48 : 32) == 0 && (fuse || p.N <= 32 || p.N >= 48)) {
#define HALO_LAUNCH(EPI_, CS_, NWB_, SWZ_, BN_, ...)                                                                        \
    {                                                                                                              \
        constexpr int lds_ = HaloCfg<CS_, NWB_, SWZ_, BN_, ##__VA_ARGS__>::LDS;                                             \
        static ScailDeviceOnce attr_;                                                                                 \
        if (attr_.need()) {                                                                                              \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<EPI_, CS_, NWB_, SWZ_, BN_, ##__VA_ARGS__>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) {             \
                scail_set_error("conv3d: hipFuncSetAttribute failed");                                             \
                return 2;                                                                                          \
            }                                                                                                      \
            attr_.done();                                                                                          \
        }                                                                                                          \
        hipLaunchKernelGGL((conv_halo_kernel<EPI_, CS_, NWB_, SWZ_, BN_, ##__VA_ARGS__>), dim3((unsigned)tiles), dim3(256), lds_, (hipStream_t)stream, p); \
    }
        const int hbn = p.N <= 32 ? 32 : 96;
        const int nf = ((g_conv_halo == 4 || g_conv_halo == 5 || fuse) && hbn == 96 && p.To >= 2) ? 2 : 1;   // output frames per workgroup (an odd last frame pair wastes half a tile: 1 / 81)
        if (fuse) {                                                        // fused RMS_norm + SiLU epilogue: one N tile
            SCAIL_REQUIRE(p.N <= 96 && resid == nullptr && (reinterpret_cast<uintptr_t>(gamma) & 15) == 0,
                          "conv + norm fusion needs N <= 96, no residual, 16-byte aligned gamma");
            const int64_t tiles = (int64_t)((p.To + nf - 1) / nf) * ((p.Ho + HT_TH - 1) / HT_TH) * ((p.Wo + HT_TW - 1) / HT_TW);
            SCAIL_REQUIRE(tiles < (1ll << 31), "too many tiles");
            if (hbn == 32) HALO_LAUNCH(4, 32, 1, false, 32) else if (nf == 2) HALO_LAUNCH(4, 32, 1, false, 96, 2) else HALO_LAUNCH(4, 32, 1, false, 96)
            return scail_check_launch("conv3d_cl_norm");
        }
        const int64_t tiles = (int64_t)((p.To + nf - 1) / nf) * ((p.Ho + HT_TH - 1) / HT_TH) * ((p.Wo + HT_TW - 1) / HT_TW) * ((p.N + hbn - 1) / hbn);
        SCAIL_REQUIRE(tiles < (1ll << 31), "too many tiles");
        if (hbn == 32) {
            if (resid != nullptr) HALO_LAUNCH(3, 32, 1, false, 32) else HALO_LAUNCH(0, 32, 1, false, 32)
#ifdef SCAIL_ABLATIONS
        } else if (nf == 2 && g_conv_halo == 5) {                          // fragment prefetch one tap ahead (A/B)
            if (resid != nullptr) HALO_LAUNCH(3, 32, 1, false, 96, 2, 3, false, 1) else HALO_LAUNCH(0, 32, 1, false, 96, 2, 3, false, 1)
#endif
        } else if (nf == 2) {
            if (resid != nullptr) HALO_LAUNCH(3, 32, 1, false, 96, 2) else HALO_LAUNCH(0, 32, 1, false, 96, 2)
#ifdef SCAIL_ABLATIONS
        } else if (g_conv_halo == 1) {
            if (resid != nullptr) HALO_LAUNCH(3, 48, 2, false, 96) else HALO_LAUNCH(0, 48, 2, false, 96)
        } else if (g_conv_halo == 2) {
            if (resid != nullptr) HALO_LAUNCH(3, 32, 2, true, 96) else HALO_LAUNCH(0, 32, 2, true, 96)
#endif
        } else {
            if (resid != nullptr) HALO_LAUNCH(3, 32, 1, false, 96) else HALO_LAUNCH(0, 32, 1, false, 96)
        }
        return scail_check_launch("conv3d_cl");
    }
    SCAIL_REQUIRE(gamma == nullptr, "conv + norm fusion covers the 3x3x3 stride-1 convolutions with Cin % 32 == 0 and N <= 96 only");
    // the 1x3x3 convolution behind the nearest 2x upsample (Resample 'upsample2d/3d', wan_vae.py:110-121): halo tile with
    // the upsampling folded into the patch load, two output frames per workgroup
    if (g_conv_halo == 4 && p.kt == 1 && p.kh == 3 && p.kw == 3 && p.st == 1 && p.sh == 1 && p.sw == 1 && p.ups &&
        p.pt == 0 && p.ph == 1 && p.pw == 1 && p.Ho == 2 * p.Hi && p.Wo == 2 * p.Wi && p.To == p.Ti && p.To >= 2 &&
        p.Cin % 32 == 0 && p.N >= 48) {
        const int64_t tiles = (int64_t)((p.To + 1) / 2) * ((p.Ho + HT_TH - 1) / HT_TH) * ((p.Wo + HT_TW - 1) / HT_TW) * ((p.N + 95) / 96);
        SCAIL_REQUIRE(tiles < (1ll << 31), "too many tiles");
        if (resid != nullptr) HALO_LAUNCH(3, 32, 1, false, 96, 2, 1, true) else HALO_LAUNCH(0, 32, 1, false, 96, 2, 1, true)
        return scail_check_launch("conv3d_cl");
    }
    // Resample's spatial downsampling: 1 x 3 x 3, stride (1, 2, 2), zero padding behind the last row / column only (conv_s2_kernel)
    if (g_conv_s2 && !rn && resid == nullptr && p.kt == 1 && p.kh == 3 && p.kw == 3 && p.st == 1 && p.sh == 2 && p.sw == 2 && !p.ups &&
        p.pt == 0 && p.ph == 0 && p.pw == 0 && p.To == p.Ti && p.Cin % S2_CS == 0 && p.N % 96 == 0 &&
        2 * (int64_t)p.Hi * p.Wi * p.Cin < (1ll << 31) && (int64_t)p.N * p.Kpad < (1ll << 31)) {
        const int nf = (g_conv_s2 == 2 && p.To >= 2) ? 2 : 1;
        const int64_t vtiles = (int64_t)((p.To + nf - 1) / nf) * ((p.Ho + S2_TH - 1) / S2_TH) * ((p.Wo + S2_TW - 1) / S2_TW);
        const int64_t grid = (vtiles + 7) / 8 * 8 * (p.N / 96);
        SCAIL_REQUIRE(grid < (1ll << 31), "too many tiles");
#define S2_LAUNCH(NF_)                                                                                                              \
    {                                                                                                                               \
        static ScailDeviceOnce attr_;                                                                                               \
        if (attr_.need()) {                                                                                                         \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_kernel<NF_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    S2Cfg<NF_>::LDS) != hipSuccess) {                                                               \
                scail_set_error("conv3d: hipFuncSetAttribute failed");                                                              \
                return 2;                                                                                                           \
            }                                                                                                                       \
            attr_.done();                                                                                                           \
        }                                                                                                                           \
        hipLaunchKernelGGL((conv_s2_kernel<NF_>), dim3((unsigned)grid), dim3(256), S2Cfg<NF_>::LDS, (hipStream_t)stream, p, (int)vtiles); \
    }
        if (nf == 2) S2_LAUNCH(2) else S2_LAUNCH(1)
#undef S2_LAUNCH
        return scail_check_launch("conv3d_cl");
    }
    // HBM-bound shapes: the direct-gather kernel (comment above conv_direct_kernel); N = 384 as two launches of 192 channels
    {
        const int ksteps = (p.Ktrue + 15) / 16;
        const int nsplit = p.N == 384 ? 2 : 1, nn = p.N / nsplit, nb = nn / 32;
        // (instantiated: up to 6 k-steps with 32 / 64 / 96 / 128 / 192 channels -- the 1x1x1 convolutions of up to 96 input channels --, up to
        // 14 k-steps with 96 channels -- the stem; 14 x 192 and 20 x 96 were built and spill 46 / 22 registers: those shapes stay on the gather kernel)
        const bool shape = ((ksteps <= 6 && (nb == 1 || nb == 2 || nb == 3 || nb == 4 || nb == 6)) || (ksteps <= 14 && nb == 3)) &&
                           p.kt <= 30 && p.kh <= 31 && p.kw <= 31 && p.Cin < 65536 &&
                           (int64_t)(p.kt + 1) * p.Hi * p.Wi * p.Cin < (1ll << 31);
        const int wld = ksteps * 16 + 8;                              // padded W row: 16 consecutive rows start in distinct 16-byte bank groups
        const int lds = (nn * wld + 8 * 32 * CD_STRIP_LD) * 2 + 2 * nn * 4;        // W, the waves' strips, bias + gamma
        if (g_conv_direct && !p.ups && resid == nullptr && p.Cin % 8 == 0 && nn % 32 == 0 && shape && lds <= 150 * 1024 && ldc % 8 == 0 &&
            (reinterpret_cast<uintptr_t>(y) & 15) == 0 && p.M >= 4096 && p.M < (1ll << 40) && p.Ho * (int64_t)p.Wo < (1ll << 31) && p.Ho * (int64_t)p.Wo >= 32) {
            const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((p.M + 255) / 256, conv4_cu_count()));
            if (rn) {       // raw output + the consumer's normalised input (scail_conv3d_cl_resid_norm checked conv_direct_dual_eligible)
                SCAIL_REQUIRE(nb == 3 && nsplit == 1 && ksteps <= 14 && y != y_norm, "conv3d: dual-output direct kernel needs 96 channels, <= 14 k-steps");
                ConvParams q = p;
                q.gamma = next_gamma;
                q.y2 = y_norm;
                static ScailDeviceOnce attr_;
                if (attr_.need()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<14, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
                        scail_set_error("conv3d: hipFuncSetAttribute failed");
                        return 2;
                    }
                    attr_.done();
                }
                hipLaunchKernelGGL((conv_direct_kernel<14, 3, true>), dim3(grid), dim3(CD_THREADS), lds, (hipStream_t)stream, q, wld);
                return scail_check_launch("conv3d_cl_resid_norm");
            }
#define CD_LAUNCH(KS_, NB_)                                                                                                          \
    {                                                                                                                                \
        static ScailDeviceOnce attr_;                                                                                                   \
        if (attr_.need()) {                                                                                                                \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<KS_, NB_>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) { \
                scail_set_error("conv3d: hipFuncSetAttribute failed");                                                               \
                return 2;                                                                                                            \
            }                                                                                                                        \
            attr_.done();                                                                                                            \
        }                                                                                                                            \
        hipLaunchKernelGGL((conv_direct_kernel<KS_, NB_>), dim3(grid), dim3(CD_THREADS), lds, (hipStream_t)stream, q, wld);          \
    }
            for (int part = 0; part < nsplit; ++part) {
                ConvParams q = p;
                q.N = nn;
                q.w = p.w + (int64_t)part * nn * p.Kpad;
                q.bias = p.bias ? p.bias + part * nn : nullptr;
                q.y = p.y + part * nn;
                if (ksteps <= 6) {
                    if (nb == 1) CD_LAUNCH(6, 1) else if (nb == 2) CD_LAUNCH(6, 2) else if (nb == 3) CD_LAUNCH(6, 3) else if (nb == 4) CD_LAUNCH(6, 4) else CD_LAUNCH(6, 6)
                } else {
                    CD_LAUNCH(14, 3)
                }
            }
#undef CD_LAUNCH
            return scail_check_launch("conv3d_cl");
        }
    }
    SCAIL_REQUIRE(!rn, "conv3d: the dual-output form was requested for a shape no dual-output kernel covers (scail_conv3d_kernel_for(.., 2) decides)");
    // N tile: of 128 / 96 / 64 the one with the fewest padding columns (ties -> the wider tile)
    int bn = 128;
    {
        int64_t best = (p.N + 127) / 128 * 128;
        const int64_t c96 = (p.N + 95) / 96 * 96, c64 = (p.N + 63) / 64 * 64;
        if (c96 < best) { best = c96; bn = 96; }
        if (c64 < best) { best = c64; bn = 64; }
    }
#define CONV_LAUNCH(EPI_, BN_, WM_, WN_)                                                                           \
    {                                                                                                              \
        constexpr int lds_ = 2 * (CBM + BN_) * CLDT * 2;                                                           \
        static ScailDeviceOnce attr_;                                                                                 \
        if (attr_.need()) {                                                                                              \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<EPI_, BN_, WM_, WN_>),        \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) {             \
                scail_set_error("conv3d: hipFuncSetAttribute failed");                                             \
                return 2;                                                                                          \
            }                                                                                                      \
            attr_.done();                                                                                          \
        }                                                                                                          \
        const int64_t tiles = ((p.M + CBM - 1) / CBM) * ((p.N + BN_ - 1) / BN_);                                   \
        SCAIL_REQUIRE(tiles < (1ll << 31), "too many tiles");                                                      \
        hipLaunchKernelGGL((conv_igemm_kernel<EPI_, BN_, WM_, WN_>), dim3((unsigned)tiles), dim3(CONV_THREADS), lds_, \
                           (hipStream_t)stream, p);                                                                \
    }
    if (resid != nullptr) {
        if (bn == 64) CONV_LAUNCH(3, 64, 4, 1) else if (bn == 96) CONV_LAUNCH(3, 96, 4, 1) else CONV_LAUNCH(3, 128, 2, 2)
    } else {
        if (bn == 64) CONV_LAUNCH(0, 64, 4, 1) else if (bn == 96) CONV_LAUNCH(0, 96, 4, 1) else CONV_LAUNCH(0, 128, 2, 2)
    }
    return scail_check_launch("conv3d_cl");
}

extern "C" int scail_conv3d_cl(const scail_bf16* x, const scail_bf16* w, const float* bias, scail_bf16* y, int64_t ldc,
                               const scail_bf16* resid, int64_t ldr, const int32_t* geom, void* stream) {
    return conv3d_impl(x, w, bias, y, ldc, resid, ldr, nullptr, geom, stream);
}

extern "C" int scail_conv3d_cl_norm(const scail_bf16* x, const scail_bf16* w, const float* bias, scail_bf16* y, int64_t ldc,
                                    const float* gamma, const int32_t* geom, void* stream) {
    SCAIL_REQUIRE(gamma != nullptr, "null argument");
    return conv3d_impl(x, w, bias, y, ldc, nullptr, 0, gamma, geom, stream);
}

extern "C" int scail_rms_silu(const scail_bf16* x, scail_bf16* y, const float* gamma, int64_t nvox, int64_t C, int silu, void* stream);

extern "C" int scail_conv3d_cl_resid_norm(const scail_bf16* x, const scail_bf16* w, const float* bias, scail_bf16* y, scail_bf16* y_norm, int64_t ldc,
                                          const scail_bf16* resid, int64_t ldr, const float* gamma, const int32_t* geom, void* stream) {
    SCAIL_REQUIRE(x != nullptr && w != nullptr && y_norm != nullptr && gamma != nullptr && geom != nullptr, "null argument");
    SCAIL_REQUIRE(y != y_norm, "y and y_norm must be different tensors (pass y = NULL when the raw sum is not needed)");
    SCAIL_REQUIRE(resid != nullptr || y != nullptr, "without a residual the raw output is required (conv + norm alone is scail_conv3d_cl_norm)");
    if (resid == nullptr) ldr = 0;
    ConvParams p;
    conv_params(p, geom);
    SCAIL_REQUIRE(ldc == p.N && p.ot_mul == 1 && p.ot_off == 0, "scail_conv3d_cl_resid_norm needs dense outputs: ldc == N, ot_mul = 1, ot_off = 0");
    const bool aligned = ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(y_norm) | reinterpret_cast<uintptr_t>(resid) |
                           reinterpret_cast<uintptr_t>(gamma)) & 15) == 0;
    if (aligned && scail_conv3d_kernel_for(geom, ldc, ldr, 2) != 0)       // 4: a generated kernel, 2: the direct-gather kernel's dual-output form
        return conv3d_impl(x, w, bias, y, ldc, resid, ldr, nullptr, geom, stream, gamma, y_norm);
    // everything else: the two separate calls this entry point stands for
    scail_bf16* raw = y != nullptr ? y : y_norm;
    if (int rc = conv3d_impl(x, w, bias, raw, ldc, resid, ldr, nullptr, geom, stream)) return rc;
    return scail_rms_silu(raw, y_norm, gamma, p.M, p.N, 1, stream);
}

extern "C" int scail_rms_silu(const scail_bf16* x, scail_bf16* y, const float* gamma, int64_t nvox, int64_t C, int silu,
                              void* stream) {
    SCAIL_REQUIRE(C % 8 == 0 && C <= 512, "C must be a multiple of 8, <= 512");
    if (nvox == 0) return 0;
    // lanes per voxel: 3 chunks per lane when the chunk count is 3 x 2^k (96, 192, 384 channels), else 1 chunk per lane
    const int nch = (int)(C / 8);
    const bool three = nch % 3 == 0 && ((nch / 3) & (nch / 3 - 1)) == 0;
    const int lanes = three ? nch / 3 : nch;
    int lg = 0;
    while ((1 << lg) < lanes) ++lg;
    constexpr int VPT = 2;
    const int64_t threads = ((nvox + VPT - 1) / VPT) << lg;
    // persistent blocks: 8 per CU (40 registers per lane), each walking over its share of the voxel groups
    const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((threads + 255) / 256, (int64_t)conv4_cu_count() * 8)));
    if (three)
        hipLaunchKernelGGL((rms_silu_kernel<3, VPT>), grid, dim3(256), 0, (hipStream_t)stream, x, y, gamma, nvox, (int)C, lg, silu);
    else
        hipLaunchKernelGGL((rms_silu_kernel<1, VPT>), grid, dim3(256), 0, (hipStream_t)stream, x, y, gamma, nvox, (int)C, lg, silu);
    return scail_check_launch("rms_silu");
}

extern "C" int scail_softmax_rows(scail_bf16* s, int64_t ld, int64_t rows, int64_t n, float scale, void* stream) {
    SCAIL_REQUIRE(n >= 1 && n <= 32768 && ld % 8 == 0 && ld >= (n + 7) / 8 * 8 && (reinterpret_cast<uintptr_t>(s) & 15) == 0,
                  "need 1 <= n <= 32768, ld a multiple of 8 that covers ceil8(n), 16-byte aligned rows");
    if (rows == 0) return 0;
    if (n <= 8192) hipLaunchKernelGGL(softmax_rows_kernel<4>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, ld, (int)n, scale);
    else hipLaunchKernelGGL(softmax_rows_kernel<16>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, ld, (int)n, scale);
    return scail_check_launch("softmax_rows");
}

extern "C" int scail_transpose2d(const scail_bf16* in, int64_t ldi, int64_t in_bs, scail_bf16* out, int64_t ldo,
                                 int64_t out_bs, int64_t R, int64_t Cc, int64_t batch, void* stream) {
    if (R == 0 || Cc == 0 || batch == 0) return 0;
    hipLaunchKernelGGL(transpose2d_kernel, dim3((unsigned)((Cc + 63) / 64), (unsigned)((R + 63) / 64), (unsigned)batch),
                       dim3(256), 0, (hipStream_t)stream, in, ldi, in_bs, out, ldo, out_bs, (int)R, (int)Cc);
    return scail_check_launch("transpose2d");
}

extern "C" int scail_to_channels_last(const float* x, scail_bf16* y, const float* a, const float* b, int64_t C,
                                      int64_t Cpad, int64_t N, void* stream) {
    const int64_t total = N * Cpad;
    if (total == 0) return 0;
    hipLaunchKernelGGL(to_channels_last_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       y, a, b, (int)C, (int)Cpad, N);
    return scail_check_launch("to_channels_last");
}

extern "C" int scail_from_channels_last(const scail_bf16* x, int64_t ldx, float* y, const float* a, const float* b,
                                        int64_t C, int64_t N, float lo, float hi, void* stream) {
    const int64_t total = N * C;
    if (total == 0) return 0;
    hipLaunchKernelGGL(from_channels_last_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, y, a, b, (int)C, N, lo, hi);
    return scail_check_launch("from_channels_last");
}

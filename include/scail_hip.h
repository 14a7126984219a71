/*
 * scail_hip.h -- C ABI of libscail_hip.so, the MI355X (gfx950) implementation of the SCAIL
 * video-DiT sampling hot path.
 *
 * The reference (zai-org/SCAIL) has no C/FFI boundary: its plugin seam is Python (config-named
 * classes + SAT mixin hooks).  This header is the boundary a maintainer
 * binds instead of the torch calls the reference makes at those seams; every entry point cites
 * the reference code it replaces (paths relative to the reference root).  INTEGRATION.md shows
 * the ctypes binding and where each call plugs into the reference's hooks.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *   - caller owns every buffer; nothing is retained after the call returns (no handles yet);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no host sync inside;
 *   - activations / weights are bf16 (uint16 storage), row-major, weights in torch Linear
 *     layout [out, in] (sat/mpu/layers.py:208-210); small per-channel vectors are fp32;
 *   - every function returns 0 on success, non-zero on error; scail_last_error() returns the
 *     message of the calling thread's last error (reference convention: Python exceptions /
 *     asserts, e.g. dit_video_crossattn_sc_xc.py:1456 -- the binding raises RuntimeError).
 */
#ifndef SCAIL_HIP_H
#define SCAIL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t scail_bf16;

const char* scail_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int scail_abi_version(void);

/* ---- epilogues of scail_gemm_bf16 ---------------------------------------------------------- */
enum {
    SCAIL_EPI_BIAS = 0,       /* y = x W^T + b                                                  */
    SCAIL_EPI_GELU_TANH = 1,  /* y = gelu_tanh(x W^T + b)   (MLP up-proj, text_embedding)         */
    SCAIL_EPI_GELU_ERF = 2,   /* y = gelu(x W^T + b)        (MLPProj, dit...:31-45)               */
    SCAIL_EPI_RESID = 3       /* y = resid + gate[b,:] * (x W^T + b); gate==NULL -> ungated        */
};

/*
 * y[M,N] = epilogue( x[M,K] . W[N,K]^T + bias[N] ),  bf16 in, fp32 accumulate on MFMA, bf16 out.
 * Replaces every F.linear on the per-token path: QKV / dense (sat/model/transformer.py:66-118,
 * dit...:1065-1101), cross query/dense (dit...:1116,1200), MLP (sat/transformer_defaults.py:163-176),
 * patch embedding as a per-patch linear (dit...:99-130), final linear (dit...:826).
 *   lda/ldr/ldc: row strides in elements.  K % 64 == 0, N % 8 == 0.  bias may be NULL.
 *   SCAIL_EPI_RESID: gate is fp32 [n_batch, gate_stride] indexed gate[(m / rows_per_batch) * gate_stride + n]
 *   (the AdaLN gate of dit...:1036,1050); resid may alias y.
 */
int scail_gemm_bf16(const scail_bf16* x, int64_t lda, const scail_bf16* w, const float* bias,
                    scail_bf16* y, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                    const scail_bf16* resid, int64_t ldr, const float* gate, int64_t gate_stride,
                    int64_t rows_per_batch, void* stream);

/*
 * LayerNorm (no affine, eps) + AdaLN modulate:  y = LN(x) * (1 + scale[b]) + shift[b]
 * (sat/ops/layernorm.py:16-24 + modulate dit...:760-761, used at :1031-1032, :1045-1046, :825).
 * Output row r in [0, n_batch*rows_out) reads source row
 *   (r / rows_out) * src_rows_per_batch + src_row_offset + (r % rows_out)
 * so the final layer can normalise only the noise-token slice (dit...:771).
 * shift/scale: fp32, indexed [b * mod_stride + col].
 */
int scail_ln_modulate(const scail_bf16* x, int64_t ldx, scail_bf16* y, int64_t ldy,
                      const float* shift, const float* scale, int64_t mod_stride,
                      int64_t n_batch, int64_t rows_out, int64_t src_rows_per_batch,
                      int64_t src_row_offset, int64_t D, float eps, void* stream);

/* LayerNorm with affine weight/bias (post_cross_attention_layernorm, sat/model/transformer.py:409;
 * nn.LayerNorm in MLPProj dit...:36,40).  w, b fp32 [D]. */
int scail_layernorm_affine(const scail_bf16* x, int64_t ldx, scail_bf16* y, int64_t ldy,
                           const float* w, const float* b, int64_t rows, int64_t D, float eps,
                           void* stream);

/*
 * RMSNorm over the FULL hidden dim D (RMSNorm dit...:48-68 with hidden_size_head == D, yaml :72)
 * followed (optionally) by the interleaved-pair 3D RoPE of Rotary3DPositionEmbeddingMixin
 * (dit...:336-340, 556-557) with per-token cos/sin tables [L, head_dim/2] (fp32) built by the
 * host from the three segment rules (ref / noise / pose, dit...:525-645).
 *   x, y: [rows, D] with strides ldx / ldy (y may alias x); w fp32 [D];
 *   cos/sin == NULL -> no rotation (cross-attention q/k, dit...:1131-1142).
 *   token index for the tables = row % rows_per_batch.
 */
int scail_rmsnorm_rope(const scail_bf16* x, int64_t ldx, scail_bf16* y, int64_t ldy, const float* w,
                       const float* cos_tab, const float* sin_tab, int64_t rows,
                       int64_t rows_per_batch, int64_t D, int64_t head_dim, float eps, void* stream);

/*
 * V -> V^T staging for scail_flash_attn_bf16: v [n_batch, Lk, heads*head_dim] (row stride ldv,
 * batch stride v_batch_stride, elements) -> vt [n_batch, heads, head_dim, Lkp], Lkp = ceil64(Lk),
 * zero padded; inside every 16-key group key bits 2 and 3 are swapped so that one 16-byte LDS
 * read yields the 8 keys an MFMA 32x32x16 k-slot group consumes (see DESIGN.md "P.V operand").
 */
int scail_transpose_v(const scail_bf16* v, int64_t ldv, int64_t v_batch_stride, scail_bf16* vt,
                      int64_t n_batch, int64_t heads, int64_t head_dim, int64_t Lk, void* stream);

/*
 * Un-masked softmax attention, head_dim 128, replaces F.scaled_dot_product_attention at
 * sat/transformer_defaults.py:67-72 and the layout churn around it (dit...:1078-1100).
 *   q  [n_batch, Lq, heads, 128]   element strides (q_bs, q_rs); head h at column h*128
 *   k  [n_seg][n_batch, Lk, heads, 128]  strides (k_ss, k_bs, k_rs)
 *   vt [n_seg][n_batch, heads, 128, Lkp] from scail_transpose_v; strides (vt_ss, vt_bs); Lkp=ceil64(Lk)
 *   o  [n_batch, Lq, heads, 128]   strides (o_bs, o_rs); accumulate != 0 -> o += result
 *      (text + CLIP cross-attention sum, dit...:1197)
 * n_seg > 1: the key axis is the concatenation of n_seg equally sized segments (sequence-parallel
 * K/V all-gather); a batch stride of 0 broadcasts K/V over the batch.
 * softmax scale = `scale` (1/sqrt(128) in the reference).
 */
int scail_flash_attn_bf16(const scail_bf16* q, int64_t q_bs, int64_t q_rs,
                          const scail_bf16* k, int64_t k_ss, int64_t k_bs, int64_t k_rs,
                          const scail_bf16* vt, int64_t vt_ss, int64_t vt_bs,
                          scail_bf16* o, int64_t o_bs, int64_t o_rs,
                          int64_t n_batch, int64_t heads, int64_t Lq, int64_t Lk, int64_t n_seg,
                          float scale, int accumulate, void* stream);

/*
 * Which kernel scail_flash_attn_bf16 runs for a shape: 4 = the hand-scheduled 4-wave kernel (csrc/attn4.s, generated by
 * scail_amd/asmgen/attn4.py: one wave per SIMD, 64 query rows per wave; needs Lk % 64 == 0, Lk >= 512, no accumulate and
 * 32-bit byte offsets inside one (batch, head) slice), 8 = the 8-wave kernels of csrc/attn.hip (everything else: ragged key
 * counts, the short text / CLIP key sets, accumulate).  Host-only query; lets a caller (and the tests) see the path taken.
 */
int scail_flash_attn_kernel_for(int64_t q_rs, int64_t k_rs, int64_t o_rs, int64_t Lq, int64_t Lk, int accumulate);

/*
 * Cross attention over TWO key sets in one launch:
 *     o = bf16( bf16(softmax(scale q k1^T) v1) + softmax(scale q k2^T) v2 )
 * Replaces the text + CLIP-image cross attention of CrossAttentionLayer (dit_video_crossattn_sc_xc.py:1107-1203: two
 * attention_fn calls over the same queries, outputs added in bf16) -- independent softmax states per set, Q read once, O written
 * once (no read-modify-write).  Layouts as scail_flash_attn_bf16 with n_seg = 1: q / o row-strided token-major views with
 * heads * 128 columns, k1 / k2 (batch stride 0 = shared by the batch) with Lk1 / Lk2 >= 1 valid keys each, vt1 / vt2 =
 * scail_transpose_v images (heads, 128, ceil64(Lk)) per batch element (batch stride 0 = shared).  Meant for short key sets
 * (hundreds of keys): 128-row query blocks, two workgroups per CU.
 */
int scail_cross_attn2_bf16(const scail_bf16* q, int64_t q_bs, int64_t q_rs,
                           const scail_bf16* k1, int64_t k1_bs, int64_t k1_rs, const scail_bf16* vt1, int64_t vt1_bs, int64_t Lk1,
                           const scail_bf16* k2, int64_t k2_bs, int64_t k2_rs, const scail_bf16* vt2, int64_t vt2_bs, int64_t Lk2,
                           scail_bf16* o, int64_t o_bs, int64_t o_rs,
                           int64_t n_batch, int64_t heads, int64_t Lq, float scale, void* stream);

/* Sinusoidal timestep embedding in fp64 like sgm/modules/diffusionmodules/util.py:207-231:
 * out[b, :dim/2] = cos(t*f), out[b, dim/2:] = sin(t*f), f_j = exp(-ln(1e4) j/(dim/2)). fp32 out. */
int scail_timestep_embedding(const float* t, float* out, int64_t n, int64_t dim, void* stream);

/* y[M,N] = act_out( act_in(x[M,K]) . W[N,K]^T + b ), M <= 8, x/y fp32, W bf16, b fp32.
 * act codes: 0 none, 1 SiLU, 2 GELU-tanh.  time_embed / adaln_projection (dit...:1327-1335,1524,1555). */
int scail_small_linear(const float* x, const scail_bf16* w, const float* b, float* y, int64_t M,
                       int64_t N, int64_t K, int act_in, int act_out, void* stream);

/* out[layer, b, j] = emb[b, j] + table[layer, j]   (AdaLN tables, dit...:1025-1028 and :823). fp32. */
int scail_adaln_table(const float* emb, const float* table, float* out, int64_t n_layers,
                      int64_t n_batch, int64_t width, void* stream);

/*
 * Token assembly for the patch embedding (DiffusionTransformer.forward dit...:1457-1503 +
 * ImagePatchEmbeddingMixin dit...:99-130): writes the per-patch im2col matrix
 *   tok [n_batch, L, kpad] bf16,  L = (1+T)*(H/2)*(W/2) + T*(H/4)*(W/4), kpad >= 80 (zero padded),
 * column (c*4 + p*2 + q), c<16 latent channel, c in 16..19 mask channel (0 noise, 1 ref/pose),
 * token order [ref | noise | pose], each (t h w) row-major.
 *   x fp32 [n_batch,T,16,H,W] (rounded to bf16 as dit...:1454-1455); ref bf16 [n_ref,1,16,H,W];
 *   pose bf16 [n_pose,T,16,H/2,W/2]; n_ref/n_pose in {1, n_batch} (CFG repeat :1479-1495).
 */
int scail_patchify(const float* x, const scail_bf16* ref, const scail_bf16* pose, scail_bf16* tok,
                   int64_t n_batch, int64_t n_ref, int64_t n_pose, int64_t T, int64_t H, int64_t W,
                   int64_t kpad, void* stream);

/* unpatchify dit...:764-784: tok [n_batch, T*(H/2)*(W/2), 64] bf16 ((o p q c) columns)
 * -> out fp32 [n_batch, T, 16, H, W]. */
int scail_unpatchify(const scail_bf16* tok, float* out, int64_t n_batch, int64_t T, int64_t H,
                     int64_t W, void* stream);

/* x += dsigma * (v_u + cfg * (v_c - v_u)),  v = [v_u; v_c] fp32 [2, n]
 * (guiders.py:41-45 + sampling_utils.py:7-10 + Euler update sampling.py:960-963), all fp32. */
int scail_cfg_euler(float* x, const float* v, int64_t n, float cfg_scale, float dsigma, void* stream);


/* ---- Wan2.1 causal 3D VAE (sgm/models/wan_vae.py), channels-last (T,H,W,C) bf16 activations ------------- */

/*
 * Implicit-GEMM convolution, whole sequence in one pass:
 *   y[voxel(to,ho,wo), n] = bias[n] + sum_{dt,dh,dw,c} x[ti, hi, wi, c] * w[n, ((dt*kh+dh)*kw+dw)*Cin + c] (+ resid)
 *   ti = to*st + dt - pt,  hi = ho*sh + dh - ph,  wi = wo*sw + dw - pw   (out-of-range -> 0);
 *   ups != 0: the input is virtually nearest-exact upsampled 2x in H and W first (hi, wi index the
 *   upsampled grid, source pixel = index >> 1)                                  (Upsample + Conv2d, :76-85)
 * Covers CausalConv3d (:17-36: pt = 2*p front padding only), the stride-2 downsampling convs with
 * ZeroPad2d((0,1,0,1)) (:87-96: ph = pw = 0) and the temporal convs of Resample (:84-96).
 *   x (Ti,Hi,Wi,Cin) with Cin % 8 == 0;  w (>= N rows, Kpad) bf16, K zero-padded to a multiple of 64;
 *   output voxel index = ((to*ot_mul + ot_off)*Ho + ho)*Wo + wo, row stride ldc (resid likewise, ldr);
 *   geom = {Ti,Hi,Wi,Cin, To,Ho,Wo, kt,kh,kw, st,sh,sw, pt,ph,pw, ups, ot_mul,ot_off, N,Kpad} (21 int32, host).
 */
int scail_conv3d_cl(const scail_bf16* x, const scail_bf16* w, const float* bias, scail_bf16* y, int64_t ldc,
                    const scail_bf16* resid, int64_t ldr, const int32_t* geom, void* stream);

/*
 * The same convolution with RMS_norm + SiLU fused into the epilogue (ResidualBlock: residual.2 conv -> residual.3 RMS_norm ->
 * residual.4 SiLU, wan_vae.py:190-196): y = SiLU(RMS_norm(conv(x) + bias) * gamma), RMS_norm as in scail_rms_silu, applied
 * to the bf16-rounded convolution output.  Only the raw convolution output's round trip through HBM disappears; the
 * arithmetic is that of the two separate calls.  3x3x3, stride 1, 'same' spatial extent, Cin % 32 == 0, N <= 96 (one
 * output tile holds every channel of a voxel); anything else is rejected.  gamma fp32 [N], 16-byte aligned.
 */
int scail_conv3d_cl_norm(const scail_bf16* x, const scail_bf16* w, const float* bias, scail_bf16* y, int64_t ldc,
                         const float* gamma, const int32_t* geom, void* stream);

/* RMS_norm over channels (F.normalize * sqrt(C) * gamma, :39-54) + optional SiLU; x,y (nvox, C), gamma fp32. */
int scail_rms_silu(const scail_bf16* x, scail_bf16* y, const float* gamma, int64_t nvox, int64_t C, int silu,
                   void* stream);

/* In-place row softmax of scale * s over the first n columns of each row (AttentionBlock, :252); 1 <= n <= 32768; rows are
 * padded to a multiple of 8 columns in memory (ld % 8 == 0, ld >= ceil8(n)) and columns n .. ceil8(n) are written as zeros. */
int scail_softmax_rows(scail_bf16* s, int64_t ld, int64_t rows, int64_t n, float scale, void* stream);

/* Batched 2-D transpose (R x C, row stride ldi) -> (C x R, row stride ldo). */
int scail_transpose2d(const scail_bf16* in, int64_t ldi, int64_t in_bs, scail_bf16* out, int64_t ldo,
                      int64_t out_bs, int64_t R, int64_t C, int64_t batch, void* stream);

/* planar fp32 (C, N) -> channels-last bf16 (N, Cpad): y = x*a[c] + b[c] (a, b may be NULL), pad channels zero. */
int scail_to_channels_last(const float* x, scail_bf16* y, const float* a, const float* b, int64_t C,
                           int64_t Cpad, int64_t N, void* stream);

/* channels-last bf16 (N, ldx) -> planar fp32 (C, N): y = clamp((x + b[c]) * a[c], lo, hi). */
int scail_from_channels_last(const scail_bf16* x, int64_t ldx, float* y, const float* a, const float* b,
                             int64_t C, int64_t N, float lo, float hi, void* stream);


/* ---- conditioning encoders (UMT5-XXL text encoder, CLIP ViT-H/14 visual) ---------------- */

/*
 * Small-sequence softmax attention, head_dim <= 128, K and V of one (batch, head) resident in LDS.
 *   o = softmax(scale * q k^T + bias_tab[bucket[i, j], h] + mask) v
 * strides = {q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs} (elements, host array); head h at column h*head_dim.
 * bucket int32 [Lq, Lk] + bias_tab fp32 [n_buckets, heads]: T5 relative-position bias (umt5.py:224-268, 101-113);
 * key_mask int32 [n_batch, key_mask_bs]: 0 -> key excluded (umt5.py:106-110).  CLIP: no bias/mask, scale
 * 1/sqrt(head_dim) (clip.py:96-103).
 */
int scail_attn_small(const scail_bf16* q, const scail_bf16* k, const scail_bf16* v, scail_bf16* o,
                     const int64_t* strides, int64_t n_batch, int64_t heads, int64_t Lq, int64_t Lk, int64_t head_dim,
                     float scale, const int32_t* bucket, const float* bias_tab, const int32_t* key_mask,
                     int64_t key_mask_bs, void* stream);

/* y = a * b elementwise (gated-GELU FFN of T5, umt5.py:141). */
int scail_mul_bf16(const scail_bf16* a, const scail_bf16* b, scail_bf16* y, int64_t n, void* stream);

/* y[r,:] = x[r,:] * rowscale[r] + addrow[r % add_rows,:]  (zeroing padded text rows umt5.py:522; adding the
 * ViT position embedding clip.py:317-318).  rowscale fp32 / addrow bf16 may be NULL. */
int scail_row_affine(const scail_bf16* x, scail_bf16* y, const float* rowscale, const scail_bf16* addrow, int64_t add_rows,
                     int64_t rows, int64_t D, void* stream);

/*
 * Runtime options of the library (process-wide; set before the first launch that should see them).  Returns 0, or 1 for an
 * unknown option / out-of-range value (scail_last_error names it).
 *   "attn4"      1 (default): scail_flash_attn_bf16 uses the 4-wave kernel wherever scail_flash_attn_kernel_for says 4;
 *                0: the 8-wave kernel for every shape.
 *   "attn4_thr"  lazy-rescale threshold of the 4-wave kernel in log2 units (default 8: the running row maximum is only raised,
 *                and O / l rescaled, when a score exceeds it by more than 2^8; exact either way, 0 = rescale on every new
 *                maximum like the 8-wave kernel).  Range [0, 64].
 * Schedule A/B knobs, timing ablations and cycle probes are NOT part of this library: they live in the measurement build
 * (include/scail_hip_ablation.h, SCAIL_ABLATIONS=1 python -m scail_amd.build -> scail_amd/libscail_hip_abl.so).
 */
int scail_set_option(const char* name, int value);

/* fp32 -> bf16 (round to nearest even) and back; plumbing for boundary tensors. */
int scail_f32_to_bf16(const float* x, scail_bf16* y, int64_t n, void* stream);
int scail_bf16_to_f32(const scail_bf16* x, float* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCAIL_HIP_H */

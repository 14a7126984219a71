/*
 * scail_vae.h -- seam B4 of SURVEY.md section 8b in C: the Wan2.1 causal 3D VAE (reference sgm/models/wan_vae.py,
 * `WanVAE_.encode` :516-542 / `.decode` :544-568) as two calls into libscail_hip.so, composed in C++ from the
 * operator entry points of scail_hip.h (scail_conv3d_cl, scail_rms_silu, scail_gemm_bf16, scail_softmax_rows, ...).
 * Whole-sequence execution (no 1/4/4-frame chunking, no feature caches): see DESIGN.md section 4.4 for the temporal
 * rules that make it equal to the reference's streamed computation.
 *
 * Ownership: the caller owns weights, inputs, outputs and the workspace; the handle copies only the pointer tables.
 * Weights are in the kernel layout (`scail_amd.ops.prep_conv_weight`: [Cout_pad8][Kpad] bf16 with k = tap * Cin_pad + c,
 * bias fp32 [Cout_pad8]).  Everything is enqueued on the stream passed last; nothing synchronises.
 */
#ifndef SCAIL_VAE_H
#define SCAIL_VAE_H

#include "scail_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scail_conv_w {       /* one prepared convolution; w == NULL means "absent" (e.g. identity shortcut) */
    const scail_bf16* w;
    const float* b;
    int32_t Cin, N, Kpad, kt, kh, kw;
} scail_conv_w;

typedef struct scail_vae_res {      /* ResidualBlock, wan_vae.py:180-218: RMS-SiLU-conv-RMS-SiLU-conv (+ 1x1x1 shortcut) */
    const float* gamma0;
    scail_conv_w conv2;
    const float* gamma3;
    scail_conv_w conv6;
    scail_conv_w shortcut;
} scail_vae_res;

typedef struct scail_vae_attn {     /* AttentionBlock, wan_vae.py:221-262: single head over the H*W tokens of each frame */
    const float* gamma;
    const scail_bf16 *q_w, *k_w, *v_w, *proj_w;     /* [C][C] */
    const float *q_b, *k_b, *v_b, *proj_b;
    int32_t C;
} scail_vae_attn;

typedef struct scail_vae_stage {    /* one entry of Encoder3d.downsamples / Decoder3d.upsamples */
    int32_t kind;                   /* 0 ResidualBlock, 1 Resample down (stride-2 conv [+ temporal stride-2 conv]), 2 Resample up */
    int32_t temporal;
    scail_vae_res res;              /* kind 0 */
    scail_conv_w resample;          /* kind 1: 3x3 stride 2;  kind 2: 3x3 behind the 2x nearest upsample */
    scail_conv_w time_conv;         /* kind 1, temporal: 3x1x1 stride (2,1,1) */
    scail_conv_w time_conv0, time_conv1;   /* kind 2, temporal: the two output halves of the C -> 2C 3x1x1 conv */
} scail_vae_stage;

typedef struct scail_vae_weights {
    int32_t z_dim;
    /* encoder */
    scail_conv_w enc_conv1;
    const scail_vae_stage* enc; int32_t n_enc;
    scail_vae_res enc_mid0; scail_vae_attn enc_attn; scail_vae_res enc_mid2;
    const float* enc_head_gamma; scail_conv_w enc_head;
    scail_conv_w conv1;             /* 1x1x1 on the 2z head output */
    const float* enc_scale;         /* fp32 [z]: 1 / std  (latent normalisation, wan_vae.py:630-640) */
    const float* enc_shift;         /* fp32 [z]: -mean */
    /* decoder */
    const float* dec_scale;         /* fp32 [z]: std */
    const float* dec_shift;         /* fp32 [z]: mean */
    scail_conv_w conv2, dec_conv1;
    scail_vae_res dec_mid0; scail_vae_attn dec_attn; scail_vae_res dec_mid2;
    const scail_vae_stage* dec; int32_t n_dec;
    const float* dec_head_gamma; scail_conv_w dec_head;
} scail_vae_weights;

typedef struct scail_vae scail_vae;

int scail_vae_create(const scail_vae_weights* w, scail_vae** out);
void scail_vae_destroy(scail_vae* h);

/* Debug seam (what a forward hook on the reference's modules gives its maintainers): after every operator launch that completes a whole
 * activation -- convolutions, RMS_norm(+SiLU) passes, the mid-block attention -- `fn(user, index, op, data, T, H, W, C)` is called on the
 * host with the channels-last bf16 tensor [T][H][W][C] inside the workspace that the launch writes.  The launch is only ENQUEUED at that
 * point: the callback must order itself after the stream (enqueue its own work on the same stream, or synchronise) before reading, and
 * must not write.  The call sequence is the launch order of scail_amd/wan_vae.py's layer-by-layer path (tools/vae_exec_vs_layers.py
 * walks both); a launch with two outputs (scail_conv3d_cl_resid_norm: raw sum + the next consumer's normalised input) calls fn once per output
 * ("conv", then "conv_resid_norm").  fn == NULL switches it off (the default; nothing is called, nothing synchronises). */
typedef void (*scail_vae_trace_fn)(void* user, int index, const char* op, const void* data, int64_t T, int64_t H, int64_t W, int64_t C);
int scail_vae_set_trace(scail_vae* h, scail_vae_trace_fn fn, void* user);

/* Device workspace for a clip of T frames of H x W pixels (T = 1 + 4n, H and W multiples of 8); the same size serves decode. */
int64_t scail_vae_workspace_bytes(const scail_vae* h, int64_t T, int64_t H, int64_t W);

/* video fp32 [3, T, H, W] in [-1, 1]  ->  latent fp32 [z, 1 + (T-1)/4, H/8, W/8] (normalised mean; WanVAE_.encode). */
int scail_vae_encode(scail_vae* h, const float* video, float* latent, int64_t T, int64_t H, int64_t W,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* latent fp32 [z, Tl, hl, wl]  ->  video fp32 [3, 1 + 4 (Tl - 1), 8 hl, 8 wl], NOT clamped (WanVAE_.decode). */
int scail_vae_decode(scail_vae* h, const float* latent, float* video, int64_t Tl, int64_t hl, int64_t wl,
                     void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCAIL_VAE_H */

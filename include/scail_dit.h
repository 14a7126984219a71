/*
 * scail_dit.h -- network-level entry points of libscail_hip.so (seam B1: what
 * `OpenAIWrapper.forward -> DiffusionTransformer.forward` (wrappers.py:40-45, dit_video_crossattn_sc_xc.py:1452-1587)
 * amounts to for one sampler step), composed in C++ from the operator entry points of scail_hip.h.  No torch, no
 * Python: a C program holding device pointers can run the denoising loop with this header alone.
 *
 * Ownership: the caller owns every buffer (weights, inputs, outputs, workspace, conditioning cache); the handle
 * only keeps the configuration and COPIES OF THE POINTER TABLES (not of the weights).  All work is enqueued on the
 * hipStream_t passed last; there is no host synchronisation, so a step can be stream-captured into a hipGraph after
 * one warm-up call.  Return 0 / non-zero + scail_last_error().  Sequence-parallel ranks run the SAME executor through
 * scail_dit_step_sp / scail_dit_block_sp below: the kernels of a block are enqueued here and only the collectives of the per-layer
 * exchange are handed to the host through a callback (RCCL through torch.distributed in scail_amd/parallel.py; a C host calls
 * ncclAllToAll / ncclAllGather there).
 */
#ifndef SCAIL_DIT_H
#define SCAIL_DIT_H

#include "scail_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scail_dit_config {
    int32_t hidden_size;       /* D, multiple of 128 */
    int32_t num_heads;         /* D / 128 */
    int32_t inner_hidden_size; /* FF */
    int32_t num_layers;
    int32_t text_dim;          /* 4096 for UMT5-XXL */
    int32_t clip_dim;          /* 1280 */
    int32_t time_freq_dim;     /* 256 */
    int32_t time_embed_dim;
    float layernorm_epsilon;   /* 1e-6 */
} scail_dit_config;

/* One transformer layer (reference state_dict keys in comments; W = bf16 [out,in], b / norm weights = fp32). */
typedef struct scail_dit_layer {
    const scail_bf16* qkv_w; const float* qkv_b;     /* transformer.layers.i.attention.query_key_value */
    const scail_bf16* o_w; const float* o_b;         /* ...attention.dense */
    const float* qn; const float* kn;                /* mixins.adaln_layer.query/key_layernorm_list.i.weight */
    const scail_bf16* cq_w; const float* cq_b;       /* ...cross_attention.query */
    const scail_bf16* co_w; const float* co_b;       /* ...cross_attention.dense */
    const float* cqn;                                /* mixins.adaln_layer.cross_query_layernorm_list.i.weight */
    const float* ln_w; const float* ln_b;            /* ...post_cross_attention_layernorm */
    const scail_bf16* w1; const float* b1;           /* ...mlp.dense_h_to_4h */
    const scail_bf16* w2; const float* b2;           /* ...mlp.dense_4h_to_h */
} scail_dit_layer;

typedef struct scail_dit_weights {
    const scail_bf16* patch_w; const float* patch_b; /* mixins.patch_embed.proj, [D,128] (80 real columns, zero padded) */
    const scail_bf16* pose_w; const float* pose_b;   /* mixins.patch_embed.proj_pose, same layout */
    const scail_bf16* time0_w; const float* time0_b; /* time_embed.0 [Dt, freq] */
    const scail_bf16* time2_w; const float* time2_b; /* time_embed.2 [Dt, Dt] */
    const scail_bf16* adaln_w; const float* adaln_b; /* adaln_projection.1 [6D, Dt] */
    const float* adaln_tables;                       /* mixins.adaln_layer.adaLN_modulations, [layers, 6D] */
    const float* final_table;                        /* mixins.final_layer.adaLN_modulation, [2D] */
    const scail_bf16* final_w; const float* final_b; /* mixins.final_layer.linear [64, D] */
    const scail_dit_layer* layers;                   /* [num_layers] */
} scail_dit_weights;

/* Step-invariant conditioning (text / CLIP keys and transposed values of every layer), produced by the host once per
 * request (scail_amd.dit.DiffusionTransformer._conditioning; dit...:1505-1515, 1116-1142). */
typedef struct scail_dit_cond {
    const scail_bf16* k_text;   /* [layers, B, Lt, D] */
    const scail_bf16* vt_text;  /* [layers, B, heads, 128, ceil64(Lt)] */
    const scail_bf16* k_clip;   /* [layers, Bc, Lc, D], Bc in {1, B} */
    const scail_bf16* vt_clip;  /* [layers, Bc, heads, 128, ceil64(Lc)] */
    int64_t Lt, Lc, Bc;
} scail_dit_cond;

typedef struct scail_dit scail_dit;

int scail_dit_create(const scail_dit_config* cfg, const scail_dit_weights* w, scail_dit** out);
void scail_dit_destroy(scail_dit* h);

/* Bytes of caller-provided device workspace one step needs for a (B, T, H, W) latent batch. */
int64_t scail_dit_workspace_bytes(const scail_dit* h, int64_t B, int64_t T, int64_t H, int64_t W);

/*
 * One network evaluation: out[B,T,16,H,W] fp32 = DiT(x[B,T,16,H,W] fp32, timesteps[B] fp32 (= 1000 sigma), cond, ref, pose).
 *   ref bf16 [n_ref,1,16,H,W], pose bf16 [n_pose,T,16,H/2,W/2] (n_* in {1, B}: CFG repeat, dit...:1479-1495);
 *   rope_cos / rope_sin fp32 [L, 64] per-token pair tables for L = (1+T)(H/2)(W/2) + T(H/4)(W/4) tokens
 *   (scail_amd/rope.py; Rotary3DPositionEmbeddingMixin dit...:382-757);  workspace: scail_dit_workspace_bytes().
 *   flags: 0, or SCAIL_DIT_CFG_PAIR -- the caller states that this is the classifier-free-guidance pair of a sampler step
 *   (VanillaCFG.prepare_inputs, guiders.py:41-57: x = cat([x] * 2), s = cat([s] * 2)): B == 2, n_ref == n_pose == 1, and x[1] == x[0],
 *   timesteps[1] == timesteps[0]; only the conditioning differs.  The two elements' hidden states are then equal until the first cross
 *   attention of layer 0 (dit...:1009-1042), so the patch embedding and layer 0's LayerNorm -> QKV -> norm / RoPE -> self-attention ->
 *   out-projection run ONCE and are copied to element 1 (0.94 % of the 14B step).  Results are bit-identical to flags = 0 on such
 *   inputs (every kernel computes a row from that row alone); x[1] is not read.  The flag is a statement about the inputs, not
 *   checked on the device.  The last layer likewise evaluates only the rows the final layer reads (the noise tokens) past its K / V.
 */
#define SCAIL_DIT_CFG_PAIR 1u
int scail_dit_step(scail_dit* h, const float* x, const float* timesteps, const scail_dit_cond* cond,
                   const scail_bf16* ref, int64_t n_ref, const scail_bf16* pose, int64_t n_pose,
                   const float* rope_cos, const float* rope_sin, float* out,
                   int64_t B, int64_t T, int64_t H, int64_t W, uint32_t flags, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Seam B2 (the SAT hook `layer_forward`, dit...:1009-1051): ONE transformer block, in place on caller-owned hidden states
 * hidden bf16 [B, Ltok, D].  mod fp32 [B, 6D] = adaLN embedding + this layer's table (shift_a|scale_a|gate_a|shift_m|scale_m|gate_m);
 * cond / rope tables as for scail_dit_step (cond is indexed by `layer`).
 */
int64_t scail_dit_block_workspace_bytes(const scail_dit* h, int64_t B, int64_t Ltok);
int scail_dit_block(scail_dit* h, int64_t layer, scail_bf16* hidden, const float* mod, const scail_dit_cond* cond,
                    const float* rope_cos, const float* rope_sin, int64_t B, int64_t Ltok,
                    void* workspace, int64_t workspace_bytes, void* stream);

/*
 * ---- sequence-parallel execution (reference: DeepSpeed-Ulysses, sat/mpu/ulysses_attn_layer.py:41-110, all_to_all.py:15-108,
 * with the latent split and rank-shifted RoPE of diffusion_video.py:495-585 and dit...:1578-1585) ----------------------------------------
 * Each rank holds an aligned token slab (B, Ltok, D) of [ref | noise | pose] tokens.  A block is per-token except for the self-attention,
 * whose exchange comes in two modes:
 *   SCAIL_SP_ULYSSES    q, k, v head <-> sequence all-to-all after norm + RoPE -- ONE collective for the three (the reference: three,
 *                       ulysses_attn_layer.py:65-80) --, attention over the FULL sequence for heads / ranks heads, one all-to-all back
 *                       (heads % ranks == 0).  Buffers (bf16, caller-owned, contiguous):
 *                         send, recv  [B][ranks][Ltok][3 * D / ranks] (send: one message per destination rank, q | k | v of that rank's heads side
 *                                                                       by side in a row; recv: per source rank, i.e. one (ranks * Ltok, 3 D / ranks)
 *                                                                       matrix in rank-major token order whose column thirds are q, k, v)
 *                         ofull, back [B][ranks][Ltok][D / ranks]      (attention output of all tokens for my heads; my tokens for all head groups)
 *   SCAIL_SP_ALLGATHER  ONE exchange: all-gather of the post-RoPE K rows and of the V rows as ONE collective.  Buffers:
 *                         send [B][Ltok][2 D] (k | v side by side),  recv [B][ranks][Ltok][2 D];  ofull / back unused (NULL).
 * The executor enqueues every kernel and calls `exchange` where a collective has to be started or awaited:
 *   SCAIL_SP_FWD_START   element b's send buffers are complete on `stream` (stream order): start its forward collectives
 *                        (ulysses: all-to-all send[b] -> recv[b] with equal splits; allgather: all-gather send[b] -> recv[b]): 2 collectives per
 *                        element and layer in ulysses mode (this one + the way back), 1 in all-gather mode
 *   SCAIL_SP_FWD_WAIT    make `stream` wait for them (the kernels enqueued next read recv[b])
 *   SCAIL_SP_BACK_START  ulysses: ofull[b] is complete on `stream`: start the all-to-all ofull[b] -> back[b]
 *   SCAIL_SP_BACK_WAIT   make `stream` wait for it
 * Per layer the order of the calls is START(0), START(1), .., WAIT(0), [BACK_START(0)], WAIT(1), [BACK_START(1)], .., [BACK_WAIT(0), ..] on every
 * rank, so collectives are enqueued in the same order everywhere (with SCAIL_DIT_CFG_PAIR layer 0 exchanges element 0 only).  A non-zero
 * return aborts the step (status 3): the side streams are still joined back into `stream`, so work already enqueued stays ordered before
 * whatever the caller enqueues next, but collectives the peers started are left unmatched -- after a status 3 the caller must tear down
 * (abort) the communicator.  The callback runs on the calling thread, between launches; it must not synchronise the device if the step is
 * to stay asynchronous.
 * side_stream: both NULL = everything on `stream`.  Two streams: the two CFG elements' launches of the ulysses exchange section go to
 * side_stream[b & 1] (forked from / joined to `stream` with events inside the call): an element's rank-sized launches leave a partial last
 * round of the chip that the other element's kernels fill (pays up to 4 ranks).
 */
#define SCAIL_SP_ALLGATHER 0
#define SCAIL_SP_ULYSSES 1
#define SCAIL_SP_FWD_START 0
#define SCAIL_SP_FWD_WAIT 1
#define SCAIL_SP_BACK_START 2
#define SCAIL_SP_BACK_WAIT 3
typedef int (*scail_sp_exchange_fn)(void* user, int32_t op, int32_t layer, int32_t element, void* stream);
typedef struct scail_dit_sp {
    int32_t ranks;              /* group size, >= 2 */
    int32_t mode;               /* SCAIL_SP_ALLGATHER | SCAIL_SP_ULYSSES */
    scail_bf16* send;
    scail_bf16* recv;
    scail_bf16* ofull;
    scail_bf16* back;
    scail_sp_exchange_fn exchange;
    void* user;
    void* side_stream[2];
} scail_dit_sp;

/* One network evaluation on this rank's latent slab x [B,T,16,H,W] (H or W = the full extent / ranks; ref / pose sliced alike; rope tables
 * of the slab's tokens with the rank's window shift, scail_amd/rope.py): scail_dit_step with the self-attention exchanged as above.
 * out = this rank's slab of the result.  workspace >= scail_dit_sp_workspace_bytes (the V^T staging buffer holds ALL ranks' keys).
 * flags as for scail_dit_step.  The last layer evaluates the rank's noise rows only past its K / V exchange (all-gather: their queries too;
 * ulysses: the full-sequence attention keeps every query, its rows are rank-major). */
int64_t scail_dit_sp_workspace_bytes(const scail_dit* h, int32_t mode, int32_t ranks, int64_t B, int64_t T, int64_t H, int64_t W);
int scail_dit_step_sp(scail_dit* h, const float* x, const float* timesteps, const scail_dit_cond* cond,
                      const scail_bf16* ref, int64_t n_ref, const scail_bf16* pose, int64_t n_pose,
                      const float* rope_cos, const float* rope_sin, float* out,
                      int64_t B, int64_t T, int64_t H, int64_t W, const scail_dit_sp* sp, uint32_t flags, void* workspace, int64_t workspace_bytes,
                      void* stream);

/* Seam B2 for a sequence-parallel rank: ONE transformer block in place on this rank's hidden [B, Ltok, D] (scail_dit_block + the exchange). */
int64_t scail_dit_block_sp_workspace_bytes(const scail_dit* h, int32_t mode, int32_t ranks, int64_t B, int64_t Ltok);
int scail_dit_block_sp(scail_dit* h, int64_t layer, scail_bf16* hidden, const float* mod, const scail_dit_cond* cond,
                       const float* rope_cos, const float* rope_sin, int64_t B, int64_t Ltok, const scail_dit_sp* sp,
                       void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Timing of the executor's own launches with HIP events recorded on the launch stream (what bench.py's `roofline` objects are computed
 * from: the timed thing is the product path itself, not an instrumented copy).  scail_dit_profile(h, 1): every following scail_dit_step
 * / scail_dit_block / scail_dit_sample brackets each launch of the categories below with an event pair (events are pooled in the
 * handle; counters restart); scail_dit_profile(h, 0) stops.  scail_dit_profile_read waits for the recorded events and returns the summed
 * kernel time in ms and the number of launches of one category since the last enable.  With side streams (scail_dit_sp.side_stream) the
 * two elements' launches overlap, so the sum of a category is NOT wall-exclusive there.  Off by default; do not enable inside a stream
 * capture (event records are not capturable into a replayable graph with readable timings).
 */
#define SCAIL_DIT_PROF_SELF_ATTN 0   /* scail_flash_attn_bf16 of the self-attention (dit...:1058-1105) */
#define SCAIL_DIT_PROF_GEMM 1        /* the six per-token GEMMs of a block: qkv, attention out, cross q, cross out, MLP up, MLP down */
#define SCAIL_DIT_PROF_CROSS_ATTN 2  /* scail_cross_attn2_bf16 (dit...:1107-1203) */
/* sequence-parallel steps: the EXPOSED part of the layer exchange -- the event pair brackets nothing but the launch stream's wait for
 * the collective (SCAIL_SP_FWD_WAIT / SCAIL_SP_BACK_WAIT), so its time is what the exchange adds to the step (0 when the collective
 * finished under the other CFG element's kernels); launches = waits.  An N-rank bench line carries both (bench.py `exchange_exposed`). */
#define SCAIL_DIT_PROF_XCH_FWD_WAIT 3
#define SCAIL_DIT_PROF_XCH_BACK_WAIT 4
/* not a time: `launches` = workgroups of the profiled self-attention launches that RESTARTED (the optimistic pass of scail_attn4_m16f
 * overflowed, scail_hip.h scail_flash_attn_count_restarts; a restarted workgroup costs about twice), ms_total = 0.  0 on random data;
 * on real weights it says whether the kernel's measured rate holds (bench.py `roofline.attn_restarts_per_launch`). */
#define SCAIL_DIT_PROF_ATTN_RESTARTS 5
int scail_dit_profile(scail_dit* h, int enable);
int scail_dit_profile_read(scail_dit* h, int category, double* ms_total, int64_t* launches);

/*
 * The whole Euler sampling loop of RFSampler (sampling.py:920-982) with VanillaCFG (guiders.py:41-57) for one request:
 *   for i < n_steps:  v = DiT([x; x], timesteps[i], cond [uncond | cond], ref, pose);  x += dsigma[i] (v_u + cfg (v_c - v_u))
 * x fp32 [1,T,16,H,W] in / out (device); timesteps DEVICE fp32 [n_steps][2] (= 1000 sigma_i, twice); dsigma HOST fp32
 * [n_steps] (= sigma_{i+1} - sigma_i); cond holds the batch-2 conditioning (index 0 = uncond, 1 = cond).
 * workspace >= scail_dit_sample_workspace_bytes(h, T, H, W).  Nothing synchronises; the call returns once all steps are enqueued.
 */
int64_t scail_dit_sample_workspace_bytes(const scail_dit* h, int64_t T, int64_t H, int64_t W);
int scail_dit_sample(scail_dit* h, float* x, const float* timesteps, const float* dsigma, int64_t n_steps, float cfg_scale,
                     const scail_dit_cond* cond, const scail_bf16* ref, const scail_bf16* pose,
                     const float* rope_cos, const float* rope_sin, int64_t T, int64_t H, int64_t W,
                     void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCAIL_DIT_H */

// Network-level entry points (include/scail_dit.h): one DiT evaluation of the sampler step composed from the
// operator entry points of this library -- the C++ statement of DiffusionTransformer.forward ->
// BaseTransformer.forward -> AdaLNMixin.layer_forward (dit_video_crossattn_sc_xc.py:1452-1587, :1009-1051;
// sat/model/transformer.py:572-746) for a single sequence-parallel rank.  Host code only: every line below
// enqueues kernels on the caller's stream; nothing synchronises, so a step is hipGraph-capturable.
#include <vector>

#include "common.h"
int scail_attn4_preload();   // attn.hip
int scail_gemm4_preload();   // gemm.hip
#include "../../include/scail_dit.h"

// Optional HIP-event timing of the executor's own launches (scail_dit_profile, include/scail_dit.h): event pairs recorded on the
// launch stream around every launch of a category; the events are pooled in the handle and reused by the next enable.
constexpr int PROF_CATS = 3;     // SCAIL_DIT_PROF_SELF_ATTN / _GEMM / _CROSS_ATTN
struct ProfPool {
    std::vector<hipEvent_t> ev;  // start / stop alternating
    size_t used = 0;
};

struct scail_dit {
    scail_dit_config cfg;
    scail_dit_weights w;
    std::vector<scail_dit_layer> layers;
    bool prof = false;
    ProfPool pool[PROF_CATS];
};

namespace {

constexpr int64_t KPAD = 128;   // patch-embedding im2col width (80 real columns, zero padded to the GEMM k-tile)

inline int64_t align256(int64_t n) { return (n + 255) / 256 * 256; }

// workspace layout (bytes, 256-aligned blocks)
struct Ws {
    int64_t tok, h, xn, qkv, att, ff, vt, xf, tokout, temb, e1, emb, adaln, mod, emb2, fin, total;
};

Ws layout(const scail_dit_config& c, int64_t B, int64_t T, int64_t H, int64_t W) {
    const int64_t D = c.hidden_size, FF = c.inner_hidden_size, nh = c.num_heads;
    const int64_t hp = H / 2, wp = W / 2;
    const int64_t Lnoise = T * hp * wp, Ltok = hp * wp + Lnoise + T * (H / 4) * (W / 4);
    const int64_t Lp = (Ltok + 63) / 64 * 64;
    Ws s;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += align256(bytes); return o; };
    s.tok = take(B * Ltok * KPAD * 2);
    s.h = take(B * Ltok * D * 2);
    s.xn = take(B * Ltok * D * 2);
    s.qkv = take(B * Ltok * 3 * D * 2);
    s.att = take(B * Ltok * D * 2);
    s.ff = take(B * Ltok * FF * 2);
    s.vt = take(B * nh * 128 * Lp * 2);
    s.xf = take(B * Lnoise * D * 2);
    s.tokout = take(B * Lnoise * 64 * 2);
    s.temb = take(B * c.time_freq_dim * 4);
    s.e1 = take(B * c.time_embed_dim * 4);
    s.emb = take(B * c.time_embed_dim * 4);
    s.adaln = take(B * 6 * D * 4);
    s.mod = take((int64_t)c.num_layers * B * 6 * D * 4);
    s.emb2 = take(B * 2 * D * 4);
    s.fin = take(B * 2 * D * 4);
    s.total = off;
    return s;
}

__global__ void dup_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t rows, int64_t w) {
    // out[b, 0:w] = out[b, w:2w] = in[b, :]   (emb.repeat(1, 2) of the final-layer table add, dit...:823)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const int64_t b = i / w, j = i - b * w;
    const float v = in[i];
    out[b * 2 * w + j] = v;
    out[b * 2 * w + w + j] = v;
}

}  // namespace

#define DIT_TRY(call_)                  \
    {                                   \
        const int rc_ = (call_);        \
        if (rc_ != 0) return rc_;       \
    }

static int prof_mark(scail_dit* h, int cat, void* stream) {
    ProfPool& p = h->pool[cat];
    if (p.used == p.ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            scail_set_error("scail_dit_profile: hipEventCreate failed");
            return 2;
        }
        p.ev.push_back(e);
    }
    if (hipEventRecord(p.ev[p.used++], (hipStream_t)stream) != hipSuccess) {
        scail_set_error("scail_dit_profile: hipEventRecord failed (profiling cannot run inside a stream capture)");
        return 2;
    }
    return 0;
}
// one launch of category cat_, bracketed by an event pair when profiling is on
#define DIT_PROF(cat_, call_)                                   \
    {                                                           \
        if (h->prof) DIT_TRY(prof_mark(h, cat_, stream));       \
        DIT_TRY(call_);                                         \
        if (h->prof) DIT_TRY(prof_mark(h, cat_, stream));       \
    }

// One transformer block in place on hid (B, Ltok, D): AdaLNMixin.layer_forward, dit...:1009-1051.
//   m (B, 6D) fp32 = shift_a | scale_a | gate_a | shift_m | scale_m | gate_m of THIS layer (adaLN emb + table)
static int dit_block(scail_dit* h, int64_t i, scail_bf16* hid, const float* m, const scail_dit_cond* cond,
                     const float* rope_cos, const float* rope_sin, int64_t B, int64_t Ltok,
                     scail_bf16* xn, scail_bf16* qkv, scail_bf16* att, scail_bf16* ff, scail_bf16* vt, void* stream) {
    const scail_dit_config& c = h->cfg;
    const int64_t D = c.hidden_size, FF = c.inner_hidden_size, nh = c.num_heads;
    const float eps = c.layernorm_epsilon;
    const int64_t Lp = (Ltok + 63) / 64 * 64;
    const int64_t Ltp = (cond->Lt + 63) / 64 * 64, Lcp = (cond->Lc + 63) / 64 * 64;
    const float scale = 0.08838834764831845f;   // 1 / sqrt(128)
    const int64_t M = B * Ltok;
    scail_bf16 *q = qkv, *k = qkv + D, *v = qkv + 2 * D;   // column thirds of the fused projection, row stride 3D
    const scail_dit_layer& lw = h->layers[i];
    // -- self attention (dit...:1031-1036, :1058-1105) --
    DIT_TRY(scail_ln_modulate(hid, D, xn, D, m, m + D, 6 * D, B, Ltok, Ltok, 0, D, eps, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(xn, D, lw.qkv_w, lw.qkv_b, qkv, 3 * D, M, 3 * D, D, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
    DIT_TRY(scail_rmsnorm_rope(k, 3 * D, k, 3 * D, lw.kn, rope_cos, rope_sin, M, Ltok, D, 128, eps, stream));
    DIT_TRY(scail_transpose_v(v, 3 * D, Ltok * 3 * D, vt, B, nh, 128, Ltok, stream));
    // the queries go to the attention in log2 units (q * scale * log2 e, one rounding): its exp2 then needs no scale / shift per score
    DIT_TRY(scail_rmsnorm_rope_scaled(q, 3 * D, q, 3 * D, lw.qn, rope_cos, rope_sin, M, Ltok, D, 128, eps, scale * 1.4426950408889634f, stream));
    DIT_PROF(SCAIL_DIT_PROF_SELF_ATTN, scail_flash_attn_bf16(q, Ltok * 3 * D, 3 * D, k, 0, Ltok * 3 * D, 3 * D, vt, 0, nh * 128 * Lp, att, Ltok * D, D,
                                                             B, nh, Ltok, Ltok, 1, SCAIL_ATTN_Q_PRESCALED, 0, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(att, D, lw.o_w, lw.o_b, hid, D, M, D, D, SCAIL_EPI_RESID, hid, D, m + 2 * D, 6 * D, Ltok, stream));
    // -- cross attention: text + CLIP, ungated residual (dit...:1039-1042, :1107-1203) --
    DIT_TRY(scail_layernorm_affine(hid, D, xn, D, lw.ln_w, lw.ln_b, M, D, eps, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(xn, D, lw.cq_w, lw.cq_b, q, 3 * D, M, D, D, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
    DIT_TRY(scail_rmsnorm_rope(q, 3 * D, q, 3 * D, lw.cqn, nullptr, nullptr, M, M, D, 128, eps, stream));
    const scail_bf16* kt = cond->k_text + i * B * cond->Lt * D;
    const scail_bf16* vtt = cond->vt_text + i * B * nh * 128 * Ltp;
    const scail_bf16* kc = cond->k_clip + i * cond->Bc * cond->Lc * D;
    const scail_bf16* vtc = cond->vt_clip + i * cond->Bc * nh * 128 * Lcp;
    DIT_PROF(SCAIL_DIT_PROF_CROSS_ATTN, scail_cross_attn2_bf16(q, Ltok * 3 * D, 3 * D, kt, cond->Lt * D, D, vtt, nh * 128 * Ltp, cond->Lt,
                                                               kc, cond->Bc == 1 ? 0 : cond->Lc * D, D, vtc, cond->Bc == 1 ? 0 : nh * 128 * Lcp, cond->Lc,
                                                               att, Ltok * D, D, B, nh, Ltok, scale, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(att, D, lw.co_w, lw.co_b, hid, D, M, D, D, SCAIL_EPI_RESID, hid, D, nullptr, 0, 0, stream));
    // -- MLP (dit...:1045-1050; sat/transformer_defaults.py:163-176) --
    DIT_TRY(scail_ln_modulate(hid, D, xn, D, m + 3 * D, m + 4 * D, 6 * D, B, Ltok, Ltok, 0, D, eps, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(xn, D, lw.w1, lw.b1, ff, FF, M, FF, D, SCAIL_EPI_GELU_TANH, nullptr, 0, nullptr, 0, 0, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(ff, FF, lw.w2, lw.b2, hid, D, M, D, FF, SCAIL_EPI_RESID, hid, D, m + 5 * D, 6 * D, Ltok, stream));
    return 0;
}

// Seam B2 (SAT hook layer_forward): one block on caller-owned hidden states.  Workspace: scail_dit_block_workspace_bytes.
static int64_t block_ws(const scail_dit_config& c, int64_t B, int64_t Ltok, int64_t* off) {
    const int64_t D = c.hidden_size, FF = c.inner_hidden_size, nh = c.num_heads, Lp = (Ltok + 63) / 64 * 64;
    int64_t o = 0;
    const int64_t sizes[5] = {B * Ltok * D * 2, B * Ltok * 3 * D * 2, B * Ltok * D * 2, B * Ltok * FF * 2, B * nh * 128 * Lp * 2};
    for (int j = 0; j < 5; ++j) { off[j] = o; o += align256(sizes[j]); }
    return o;
}
extern "C" int64_t scail_dit_block_workspace_bytes(const scail_dit* h, int64_t B, int64_t Ltok) {
    if (h == nullptr || B <= 0 || Ltok <= 0) return -1;
    int64_t off[5];
    return block_ws(h->cfg, B, Ltok, off);
}
extern "C" int scail_dit_block(scail_dit* h, int64_t layer, scail_bf16* hidden, const float* mod, const scail_dit_cond* cond,
                               const float* rope_cos, const float* rope_sin, int64_t B, int64_t Ltok,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    SCAIL_REQUIRE(h != nullptr && hidden != nullptr && mod != nullptr && cond != nullptr, "null argument");
    SCAIL_REQUIRE(layer >= 0 && layer < h->cfg.num_layers && B > 0 && Ltok > 0, "bad layer / shape");
    int64_t off[5];
    const int64_t need = block_ws(h->cfg, B, Ltok, off);
    SCAIL_REQUIRE(workspace != nullptr && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                  "workspace too small or not 256-byte aligned (scail_dit_block_workspace_bytes)");
    char* base = static_cast<char*>(workspace);
    auto P = [&](int j) { return reinterpret_cast<scail_bf16*>(base + off[j]); };
    return dit_block(h, layer, hidden, mod, cond, rope_cos, rope_sin, B, Ltok, P(0), P(1), P(2), P(3), P(4), stream);
}

extern "C" int scail_dit_create(const scail_dit_config* cfg, const scail_dit_weights* w, scail_dit** out) {
    SCAIL_REQUIRE(cfg != nullptr && w != nullptr && out != nullptr, "null argument");
    SCAIL_REQUIRE(cfg->hidden_size > 0 && cfg->hidden_size % 128 == 0 && cfg->num_heads * 128 == cfg->hidden_size,
                  "hidden_size must be heads * 128");
    SCAIL_REQUIRE(cfg->num_layers > 0 && cfg->inner_hidden_size % 64 == 0 && cfg->time_embed_dim > 0 && cfg->time_freq_dim > 0,
                  "bad layer count / widths");
    SCAIL_REQUIRE(w->layers != nullptr, "weights.layers is null");
    scail_dit* h = new scail_dit;
    h->cfg = *cfg;
    h->w = *w;
    // the generated kernels live in embedded code objects loaded on first use: do it here, not inside a stream capture of the first
    // step (the per-shape tile-order table of the GEMM is still allocated by the first launch of a shape: warm up once before capturing)
    if (int rc = scail_attn4_preload()) { delete h; return rc; }
    if (int rc = scail_gemm4_preload()) { delete h; return rc; }
    h->layers.assign(w->layers, w->layers + cfg->num_layers);
    h->w.layers = h->layers.data();
    *out = h;
    return 0;
}

extern "C" void scail_dit_destroy(scail_dit* h) {
    if (h != nullptr)
        for (ProfPool& p : h->pool)
            for (hipEvent_t e : p.ev) (void)hipEventDestroy(e);
    delete h;
}

extern "C" int scail_dit_profile(scail_dit* h, int enable) {
    SCAIL_REQUIRE(h != nullptr, "null handle");
    h->prof = enable != 0;
    if (h->prof)
        for (ProfPool& p : h->pool) p.used = 0;       // a new measurement: reuse the pooled events
    return 0;
}

extern "C" int scail_dit_profile_read(scail_dit* h, int category, double* ms_total, int64_t* launches) {
    SCAIL_REQUIRE(h != nullptr && ms_total != nullptr && launches != nullptr, "null argument");
    SCAIL_REQUIRE(category >= 0 && category < PROF_CATS, "unknown category");
    ProfPool& p = h->pool[category];
    SCAIL_REQUIRE(p.used % 2 == 0, "unbalanced event pairs (a step failed between the marks)");
    double sum = 0.0;
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        if (hipEventSynchronize(p.ev[i + 1]) != hipSuccess) {
            scail_set_error("scail_dit_profile_read: hipEventSynchronize failed");
            return 2;
        }
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]) != hipSuccess) {
            scail_set_error("scail_dit_profile_read: hipEventElapsedTime failed");
            return 2;
        }
        sum += ms;
    }
    *ms_total = sum;
    *launches = (int64_t)(p.used / 2);
    return 0;
}

extern "C" int64_t scail_dit_workspace_bytes(const scail_dit* h, int64_t B, int64_t T, int64_t H, int64_t W) {
    if (h == nullptr || B <= 0 || T <= 0 || H <= 0 || W <= 0 || H % 4 != 0 || W % 4 != 0) return -1;
    return layout(h->cfg, B, T, H, W).total;
}

extern "C" int scail_dit_step(scail_dit* h, const float* x, const float* timesteps, const scail_dit_cond* cond,
                              const scail_bf16* ref, int64_t n_ref, const scail_bf16* pose, int64_t n_pose,
                              const float* rope_cos, const float* rope_sin, float* out,
                              int64_t B, int64_t T, int64_t H, int64_t W, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    SCAIL_REQUIRE(h != nullptr && cond != nullptr, "null handle / conditioning");
    SCAIL_REQUIRE(B > 0 && B <= 8 && T > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "latent batch must be 1..8, H and W multiples of 4");
    SCAIL_REQUIRE((n_ref == 1 || n_ref == B) && (n_pose == 1 || n_pose == B), "ref / pose batch must be 1 or B");
    SCAIL_REQUIRE(cond->Bc == 1 || cond->Bc == B, "clip batch must be 1 or B");
    const scail_dit_config& c = h->cfg;
    const Ws s = layout(c, B, T, H, W);
    SCAIL_REQUIRE(c.time_embed_dim == c.hidden_size, "final-layer table add needs time_embed_dim == hidden_size");
    SCAIL_REQUIRE(workspace != nullptr && workspace_bytes >= s.total, "workspace too small (scail_dit_workspace_bytes)");
    SCAIL_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");

    const int64_t D = c.hidden_size, FF = c.inner_hidden_size, nh = c.num_heads, nl = c.num_layers;
    const float eps = c.layernorm_epsilon;
    const int64_t hp = H / 2, wp = W / 2;
    const int64_t Lref = hp * wp, Lnoise = T * hp * wp, Lpose = T * (H / 4) * (W / 4);
    const int64_t Ltok = Lref + Lnoise + Lpose, Lrn = Lref + Lnoise;
    const int64_t Lp = (Ltok + 63) / 64 * 64;
    const int64_t Ltp = (cond->Lt + 63) / 64 * 64, Lcp = (cond->Lc + 63) / 64 * 64;
    char* base = static_cast<char*>(workspace);
    auto B16 = [&](int64_t off) { return reinterpret_cast<scail_bf16*>(base + off); };
    auto F32 = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
    scail_bf16 *tok = B16(s.tok), *hid = B16(s.h), *xn = B16(s.xn), *qkv = B16(s.qkv), *att = B16(s.att), *ff = B16(s.ff);
    scail_bf16 *vt = B16(s.vt), *xf = B16(s.xf), *tokout = B16(s.tokout);
    float *temb = F32(s.temb), *e1 = F32(s.e1), *emb = F32(s.emb), *adaln = F32(s.adaln), *mod = F32(s.mod);
    float *emb2 = F32(s.emb2), *fin = F32(s.fin);
    const scail_dit_weights& w = h->w;
    const float scale = 0.08838834764831845f;   // 1 / sqrt(128)

    // ---- time / AdaLN tables (dit...:1521-1555, :1025-1028, :823) ----
    DIT_TRY(scail_timestep_embedding(timesteps, temb, B, c.time_freq_dim, stream));
    DIT_TRY(scail_small_linear(temb, w.time0_w, w.time0_b, e1, B, c.time_embed_dim, c.time_freq_dim, 0, 1, stream));
    DIT_TRY(scail_small_linear(e1, w.time2_w, w.time2_b, emb, B, c.time_embed_dim, c.time_embed_dim, 0, 0, stream));
    DIT_TRY(scail_small_linear(emb, w.adaln_w, w.adaln_b, adaln, B, 6 * D, c.time_embed_dim, 1, 0, stream));
    DIT_TRY(scail_adaln_table(adaln, w.adaln_tables, mod, nl, B, 6 * D, stream));
    {
        const int64_t n = B * D;
        hipLaunchKernelGGL(dup_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, emb, emb2, B, D);
        DIT_TRY(scail_check_launch("dit_step.dup_rows"));
    }
    DIT_TRY(scail_adaln_table(emb2, w.final_table, fin, 1, B, 2 * D, stream));

    // ---- patch embedding straight into the token layout [ref | noise | pose] (dit...:99-130) ----
    DIT_TRY(scail_patchify(x, ref, pose, tok, B, n_ref, n_pose, T, H, W, KPAD, stream));
    for (int64_t b = 0; b < B; ++b) {
        DIT_TRY(scail_gemm_bf16(tok + b * Ltok * KPAD, KPAD, w.patch_w, w.patch_b, hid + b * Ltok * D, D, Lrn, D, KPAD,
                                SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
        DIT_TRY(scail_gemm_bf16(tok + (b * Ltok + Lrn) * KPAD, KPAD, w.pose_w, w.pose_b, hid + (b * Ltok + Lrn) * D, D, Lpose, D,
                                KPAD, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
    }

    for (int64_t i = 0; i < nl; ++i)
        DIT_TRY(dit_block(h, i, hid, mod + i * B * 6 * D, cond, rope_cos, rope_sin, B, Ltok, xn, qkv, att, ff, vt, stream));

    // ---- final layer on the noise tokens only + unpatchify (dit...:818-835, :764-784) ----
    DIT_TRY(scail_ln_modulate(hid, D, xf, D, fin, fin + D, 2 * D, B, Lnoise, Ltok, Lref, D, eps, stream));
    DIT_TRY(scail_gemm_bf16(xf, D, w.final_w, w.final_b, tokout, 64, B * Lnoise, 64, D, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
    DIT_TRY(scail_unpatchify(tokout, out, B, T, H, W, stream));
    return 0;
}

// ---- the sampler loop (RFSampler.__call__ + VanillaCFG, sampling.py:920-982, guiders.py:41-57) ----
extern "C" int64_t scail_dit_sample_workspace_bytes(const scail_dit* h, int64_t T, int64_t H, int64_t W) {
    const int64_t step = scail_dit_workspace_bytes(h, 2, T, H, W);
    if (step < 0) return -1;
    const int64_t n = T * 16 * H * W;
    return step + 2 * align256(2 * n * 4);     // + [x; x] and [v_u; v_c], fp32
}

extern "C" int scail_dit_sample(scail_dit* h, float* x, const float* timesteps, const float* dsigma, int64_t n_steps,
                                float cfg_scale, const scail_dit_cond* cond, const scail_bf16* ref, const scail_bf16* pose,
                                const float* rope_cos, const float* rope_sin, int64_t T, int64_t H, int64_t W,
                                void* workspace, int64_t workspace_bytes, void* stream) {
    SCAIL_REQUIRE(h != nullptr && x != nullptr && timesteps != nullptr && dsigma != nullptr && n_steps >= 0, "null argument");
    const int64_t step_bytes = scail_dit_workspace_bytes(h, 2, T, H, W);
    SCAIL_REQUIRE(step_bytes >= 0, "bad latent shape");
    const int64_t n = T * 16 * H * W, pair = align256(2 * n * 4);
    SCAIL_REQUIRE(workspace != nullptr && workspace_bytes >= step_bytes + 2 * pair, "workspace too small (scail_dit_sample_workspace_bytes)");
    char* base = static_cast<char*>(workspace);
    float* xin = reinterpret_cast<float*>(base + step_bytes);
    float* v = reinterpret_cast<float*>(base + step_bytes + pair);
    hipStream_t s = (hipStream_t)stream;
    for (int64_t i = 0; i < n_steps; ++i) {
        // torch.cat([x] * 2) of VanillaCFG.prepare_inputs (guiders.py:56)
        if (hipMemcpyAsync(xin, x, n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpyAsync(xin + n, x, n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            scail_set_error("scail_dit_sample: hipMemcpyAsync failed");
            return 2;
        }
        DIT_TRY(scail_dit_step(h, xin, timesteps + 2 * i, cond, ref, 1, pose, 1, rope_cos, rope_sin, v, 2, T, H, W, workspace,
                               step_bytes, stream));
        DIT_TRY(scail_cfg_euler(x, v, n, cfg_scale, dsigma[i], stream));
    }
    return 0;
}

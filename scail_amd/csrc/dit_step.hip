// Network-level entry points (include/scail_dit.h): one DiT evaluation of the sampler step composed from the
// operator entry points of this library -- the C++ statement of DiffusionTransformer.forward ->
// BaseTransformer.forward -> AdaLNMixin.layer_forward (dit_video_crossattn_sc_xc.py:1452-1587, :1009-1051;
// sat/model/transformer.py:572-746), for one rank or for a sequence-parallel rank (the collectives of the per-layer exchange go to
// the host through a callback).  Host code only: every line below enqueues kernels on the caller's stream; nothing synchronises, so
// a single-rank step is hipGraph-capturable.
#include <vector>

#include "common.h"
int scail_attn4_preload();   // attn.hip
void scail_attn4_rows_hint(int rows);   // attn.hip: pins the query-tile height of this thread's attention launches (0 = planned per launch)
int scail_gemm4_preload();   // gemm.hip
#include "../../include/scail_dit.h"

// Optional HIP-event timing of the executor's own launches (scail_dit_profile, include/scail_dit.h): event pairs recorded on the
// launch stream around every launch of a category; the events are pooled in the handle and reused by the next enable.
constexpr int PROF_CATS = 5;     // SCAIL_DIT_PROF_SELF_ATTN / _GEMM / _CROSS_ATTN / _XCH_FWD_WAIT / _XCH_BACK_WAIT
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
    hipEvent_t sp_ev[3] = {nullptr, nullptr, nullptr};   // fork / join events of the sequence-parallel block's side streams (created on first use)
    uint32_t* restart_ctr = nullptr;                     // device counter of restarted self-attention workgroups (allocated by the first scail_dit_profile(h, 1))
};

namespace {

constexpr int64_t KPAD = 128;   // patch-embedding im2col width (80 real columns, zero padded to the GEMM k-tile)

inline int64_t align256(int64_t n) { return (n + 255) / 256 * 256; }

// workspace layout (bytes, 256-aligned blocks)
struct Ws {
    int64_t tok, h, xn, qkv, att, ff, vt, xf, tokout, temb, e1, emb, adaln, mod, emb2, fin, total;
};

// elements of the V^T staging buffer of a (B, Ltok) block: one rank (sp_mode < 0): all heads x the local keys; ulysses: heads / ranks
// heads x ALL ranks' keys; all-gather: all heads x all ranks' keys
int64_t vt_elems(const scail_dit_config& c, int64_t B, int64_t Ltok, int sp_mode, int64_t ranks) {
    const int64_t nh = c.num_heads;
    if (sp_mode < 0) return B * nh * 128 * ((Ltok + 63) / 64 * 64);
    const int64_t Lfp = (ranks * Ltok + 63) / 64 * 64;
    return B * (sp_mode == SCAIL_SP_ULYSSES ? nh / ranks : nh) * 128 * Lfp;
}

Ws layout(const scail_dit_config& c, int64_t B, int64_t T, int64_t H, int64_t W, int sp_mode = -1, int64_t ranks = 1) {
    const int64_t D = c.hidden_size, FF = c.inner_hidden_size;
    const int64_t hp = H / 2, wp = W / 2;
    const int64_t Lnoise = T * hp * wp, Ltok = hp * wp + Lnoise + T * (H / 4) * (W / 4);
    Ws s;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += align256(bytes); return o; };
    s.tok = take(B * Ltok * KPAD * 2);
    s.h = take(B * Ltok * D * 2);
    s.xn = take(B * Ltok * D * 2);
    s.qkv = take(B * Ltok * 3 * D * 2);
    s.att = take(B * Ltok * D * 2);
    s.ff = take(B * Ltok * FF * 2);
    s.vt = take(vt_elems(c, B, Ltok, sp_mode, ranks) * 2);
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
// while profiling, the self-attention launches of this thread count their restarted workgroups into the handle's device counter
struct RestartCount {
    bool on;
    explicit RestartCount(scail_dit* h, int cat) : on(h->prof && cat == SCAIL_DIT_PROF_SELF_ATTN && h->restart_ctr != nullptr) {
        if (on) (void)scail_flash_attn_count_restarts(h->restart_ctr);
    }
    ~RestartCount() { if (on) (void)scail_flash_attn_count_restarts(nullptr); }
};
// one launch of category cat_, bracketed by an event pair when profiling is on
#define DIT_PROF(cat_, call_)                                   \
    {                                                           \
        if (h->prof) DIT_TRY(prof_mark(h, cat_, stream));       \
        RestartCount rc_guard_(h, cat_);                        \
        DIT_TRY(call_);                                         \
        if (h->prof) DIT_TRY(prof_mark(h, cat_, stream));       \
    }

// one launch of category cat_ on stream st_, bracketed by an event pair when profiling is on
#define DIT_PROF_S(cat_, st_, call_)                            \
    {                                                           \
        if (h->prof) DIT_TRY(prof_mark(h, cat_, st_));          \
        RestartCount rc_guard_(h, cat_);                        \
        DIT_TRY(call_);                                         \
        if (h->prof) DIT_TRY(prof_mark(h, cat_, st_));          \
    }

struct BlockBufs {
    scail_bf16 *xn, *qkv, *att, *ff, *vt;
};

constexpr float ATTN_SCALE = 0.08838834764831845f;              // 1 / sqrt(128)
constexpr float ATTN_LOG2_SCALE = ATTN_SCALE * 1.4426950408889634f;   // queries in log2 units (scail_flash_attn_bf16 SCAIL_ATTN_Q_PRESCALED)

// Everything of a block AFTER its self-attention, for `nb` batch elements starting at element b0 whose rows lie `rpb` apart in every
// buffer (hid / att: row stride D; q3: the first column third of a row-stride-3D buffer; xn / ff: scratch of nb * rpb rows):
// out-projection + gated residual, cross attention (text + CLIP, ungated residual), MLP (dit...:1036-1050, :1107-1203;
// sat/transformer_defaults.py:163-176).  m = the (6D) modulation row of element b0 (rows of later elements 6D apart).
// block_attn_out: the out-projection + gated residual alone; block_cross_mlp: the rest (the first point of a block at which the two CFG
// elements of a step differ: scail_dit_step's SCAIL_DIT_CFG_PAIR).
static int block_attn_out(scail_dit* h, int64_t i, scail_bf16* hid, scail_bf16* att, const float* m, int64_t nb, int64_t rpb, void* stream) {
    const int64_t D = h->cfg.hidden_size;
    const scail_dit_layer& lw = h->layers[i];
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(att, D, lw.o_w, lw.o_b, hid, D, nb * rpb, D, D, SCAIL_EPI_RESID, hid, D, m + 2 * D, 6 * D, rpb, stream));
    return 0;
}
static int block_cross_mlp(scail_dit* h, int64_t i, scail_bf16* hid, scail_bf16* att, scail_bf16* q3, scail_bf16* xn, scail_bf16* ff,
                           const float* m, const scail_dit_cond* cond, int64_t Btot, int64_t b0, int64_t nb, int64_t rpb, void* stream) {
    const scail_dit_config& c = h->cfg;
    const int64_t D = c.hidden_size, FF = c.inner_hidden_size, nh = c.num_heads;
    const float eps = c.layernorm_epsilon;
    const int64_t Ltp = (cond->Lt + 63) / 64 * 64, Lcp = (cond->Lc + 63) / 64 * 64;
    const int64_t M = nb * rpb;
    const scail_dit_layer& lw = h->layers[i];
    // -- cross attention: text + CLIP, ungated residual (dit...:1039-1042, :1107-1203) --
    DIT_TRY(scail_layernorm_affine(hid, D, xn, D, lw.ln_w, lw.ln_b, M, D, eps, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(xn, D, lw.cq_w, lw.cq_b, q3, 3 * D, M, D, D, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
    // (the queries go to the attention in log2 units, like the self-attention's: no scale / shift per score in the kernel)
    DIT_TRY(scail_rmsnorm_rope_scaled(q3, 3 * D, q3, 3 * D, lw.cqn, nullptr, nullptr, M, M, D, 128, eps, ATTN_LOG2_SCALE, stream));
    const bool shared_clip = cond->Bc == 1;
    const scail_bf16* kt = cond->k_text + (i * Btot + b0) * cond->Lt * D;
    const scail_bf16* vtt = cond->vt_text + (i * Btot + b0) * nh * 128 * Ltp;
    const scail_bf16* kc = cond->k_clip + (i * cond->Bc + (shared_clip ? 0 : b0)) * cond->Lc * D;
    const scail_bf16* vtc = cond->vt_clip + (i * cond->Bc + (shared_clip ? 0 : b0)) * nh * 128 * Lcp;
    DIT_PROF(SCAIL_DIT_PROF_CROSS_ATTN, scail_cross_attn2_bf16(q3, rpb * 3 * D, 3 * D, kt, cond->Lt * D, D, vtt, nh * 128 * Ltp, cond->Lt,
                                                               kc, shared_clip ? 0 : cond->Lc * D, D, vtc, shared_clip ? 0 : nh * 128 * Lcp, cond->Lc,
                                                               att, rpb * D, D, nb, nh, rpb, SCAIL_ATTN_Q_PRESCALED, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(att, D, lw.co_w, lw.co_b, hid, D, M, D, D, SCAIL_EPI_RESID, hid, D, nullptr, 0, 0, stream));
    // -- MLP (dit...:1045-1050; sat/transformer_defaults.py:163-176) --
    DIT_TRY(scail_ln_modulate(hid, D, xn, D, m + 3 * D, m + 4 * D, 6 * D, nb, rpb, rpb, 0, D, eps, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(xn, D, lw.w1, lw.b1, ff, FF, M, FF, D, SCAIL_EPI_GELU_TANH, nullptr, 0, nullptr, 0, 0, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(ff, FF, lw.w2, lw.b2, hid, D, M, D, FF, SCAIL_EPI_RESID, hid, D, m + 5 * D, 6 * D, rpb, stream));
    return 0;
}
// Everything after the self-attention of the rows [row0, row0 + rows) of every element (rows == Ltok: the whole block in one set of
// launches; fewer: per element, the wanted rows only).  pair: hid / att / m of element 1 do not exist yet -- the out-projection runs
// for element 0 and its result (the hidden states after the self-attention residual) is copied to element 1 before the elements part.
static int block_tail(scail_dit* h, int64_t i, scail_bf16* hid, const float* m, const scail_dit_cond* cond, int64_t B, int64_t Ltok,
                      scail_bf16* att, scail_bf16* q, scail_bf16* xn, scail_bf16* ff, int64_t row0, int64_t rows, bool pair, void* stream) {
    const int64_t D = h->cfg.hidden_size;
    if (pair) {
        DIT_TRY(block_attn_out(h, i, hid + row0 * D, att + row0 * D, m, 1, rows, stream));
        if (hipMemcpyAsync(hid + Ltok * D, hid, (size_t)Ltok * D * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
            scail_set_error("scail_dit (cfg pair): hipMemcpyAsync failed");
            return 2;
        }
    }
    if (rows == Ltok) {
        if (!pair) DIT_TRY(block_attn_out(h, i, hid, att, m, B, Ltok, stream));
        return block_cross_mlp(h, i, hid, att, q, xn, ff, m, cond, B, 0, B, Ltok, stream);
    }
    for (int64_t b = 0; b < B; ++b) {
        const int64_t r = b * Ltok + row0;
        if (!pair) DIT_TRY(block_attn_out(h, i, hid + r * D, att + r * D, m + b * 6 * D, 1, rows, stream));
        DIT_TRY(block_cross_mlp(h, i, hid + r * D, att + r * D, q + r * 3 * D, xn, ff, m + b * 6 * D, cond, B, b, 1, rows, stream));
    }
    return 0;
}

// One transformer block in place on hid (B, Ltok, D): AdaLNMixin.layer_forward, dit...:1009-1051.
//   m (B, 6D) fp32 = shift_a | scale_a | gate_a | shift_m | scale_m | gate_m of THIS layer (adaLN emb + table)
//   [row0, row0 + rows): the token rows of every batch element whose OUTPUT is wanted.  rows == Ltok: the whole block.  rows < Ltok
//   (the LAST layer of a step: only the noise tokens reach the final layer, dit...:771, :825-826): every token still contributes its
//   key and value, but the queries, the out-projection, the cross attention and the MLP run on the wanted rows only -- the other
//   rows of hid are left as they were.  Result-preserving: every kernel computes a row of its output from that row alone.
//   pair (B == 2): element 1 of hid is not filled in yet and equals element 0 up to this block's first cross attention (the CFG pair
//   of a sampler step): the self-attention part runs for element 0 only.
static int dit_block(scail_dit* h, int64_t i, scail_bf16* hid, const float* m, const scail_dit_cond* cond,
                     const float* rope_cos, const float* rope_sin, int64_t B, int64_t Ltok, const BlockBufs& bf,
                     int64_t row0, int64_t rows, bool pair, void* stream) {
    const scail_dit_config& c = h->cfg;
    const int64_t D = c.hidden_size, nh = c.num_heads;
    const float eps = c.layernorm_epsilon;
    const int64_t Lp = (Ltok + 63) / 64 * 64;
    const int64_t Bs = pair ? 1 : B;          // elements of the self-attention part
    const int64_t M = Bs * Ltok;
    scail_bf16 *qkv = bf.qkv, *q = qkv, *k = qkv + D, *v = qkv + 2 * D;   // column thirds of the fused projection, row stride 3D
    const scail_dit_layer& lw = h->layers[i];
    // -- self attention (dit...:1031-1036, :1058-1105) --
    DIT_TRY(scail_ln_modulate(hid, D, bf.xn, D, m, m + D, 6 * D, Bs, Ltok, Ltok, 0, D, eps, stream));
    DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(bf.xn, D, lw.qkv_w, lw.qkv_b, qkv, 3 * D, M, 3 * D, D, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
    DIT_TRY(scail_rmsnorm_rope(k, 3 * D, k, 3 * D, lw.kn, rope_cos, rope_sin, M, Ltok, D, 128, eps, stream));
    DIT_TRY(scail_transpose_v(v, 3 * D, Ltok * 3 * D, bf.vt, Bs, nh, 128, Ltok, stream));
    // the queries go to the attention in log2 units (q * scale * log2 e, one rounding): its exp2 then needs no scale / shift per score
    DIT_TRY(scail_rmsnorm_rope_scaled(q, 3 * D, q, 3 * D, lw.qn, rope_cos, rope_sin, M, Ltok, D, 128, eps, ATTN_LOG2_SCALE, stream));
    DIT_PROF(SCAIL_DIT_PROF_SELF_ATTN, scail_flash_attn_bf16(q + row0 * 3 * D, Ltok * 3 * D, 3 * D, k, 0, Ltok * 3 * D, 3 * D, bf.vt, 0, nh * 128 * Lp,
                                                             bf.att + row0 * D, Ltok * D, D, Bs, nh, rows, Ltok, 1, SCAIL_ATTN_Q_PRESCALED, 0, stream));
    return block_tail(h, i, hid, m, cond, B, Ltok, bf.att, q, bf.xn, bf.ff, row0, rows, pair, stream);
}

// ---- the sequence-parallel block (include/scail_dit.h "sequence-parallel execution") ----
static int sp_exchange(const scail_dit_sp* sp, int op, int64_t layer, int64_t b, void* stream) {
    const int rc = sp->exchange(sp->user, (int32_t)op, (int32_t)layer, (int32_t)b, stream);
    if (rc != 0) {
        scail_set_error("scail_dit (sequence parallel): the host's exchange callback failed (op " + std::to_string(op) + ", layer " +
                        std::to_string(layer) + ", element " + std::to_string(b) + ", status " + std::to_string(rc) + ")");
        return 3;
    }
    return 0;
}

static int sp_check(const scail_dit* h, const scail_dit_sp* sp) {
    SCAIL_REQUIRE(sp != nullptr && sp->exchange != nullptr, "null sequence-parallel descriptor / exchange callback");
    SCAIL_REQUIRE(sp->ranks >= 2 && sp->ranks <= 64, "ranks must be 2..64");
    SCAIL_REQUIRE(sp->mode == SCAIL_SP_ALLGATHER || sp->mode == SCAIL_SP_ULYSSES, "unknown exchange mode");
    SCAIL_REQUIRE(sp->send != nullptr && sp->recv != nullptr, "null send / recv buffer");
    if (sp->mode == SCAIL_SP_ULYSSES) {
        SCAIL_REQUIRE(h->cfg.num_heads % sp->ranks == 0, "ulysses needs heads divisible by the group size");
        SCAIL_REQUIRE(sp->ofull != nullptr && sp->back != nullptr, "ulysses needs the ofull / back buffers");
    }
    SCAIL_REQUIRE((sp->side_stream[0] == nullptr) == (sp->side_stream[1] == nullptr), "give both side streams or none");
    return 0;
}

// hid (B, Ltok, D) = this rank's token slab.  Kernel sequence and results are those of scail_amd.parallel's per-op path (kept as the
// cross-check: tests/test_dit_gpu.py); only the host side differs: one call per layer instead of ~30 launches through the binding.
//   [row0, row0 + rows), pair: as for dit_block.  Every token's key and value is exchanged whatever the wanted rows; with fewer rows the
//   out-projection / cross attention / MLP run on them only, and so do the queries of the all-gather mode (the ulysses attention
//   keeps all queries: its sequence is rank-major, the wanted rows would be `ranks` separate ranges).
namespace {
// Joins the side streams of the ulysses section back into `stream` -- also when the section is left early on an error: kernels and
// collectives already enqueued on the side streams stay ordered before whatever the caller enqueues on `stream` next (freeing or reusing
// the workspace and the exchange buffers included).  A status-3 abort (the host's exchange callback failed) leaves collectives unmatched
// on the peers: the caller must tear down the communicator (include/scail_dit.h).
struct SideJoin {
    scail_dit* h;
    const scail_dit_sp* sp;
    void* stream;
    int n = 0;           // side streams forked
    int join() {
        int rc = 0;
        for (int j = 0; j < n; ++j)
            if (hipEventRecord(h->sp_ev[1 + j], (hipStream_t)sp->side_stream[j]) != hipSuccess ||
                hipStreamWaitEvent((hipStream_t)stream, h->sp_ev[1 + j], 0) != hipSuccess)
                rc = 2;
        n = 0;
        return rc;
    }
    ~SideJoin() { (void)join(); }
};
}  // namespace
static int dit_block_sp(scail_dit* h, int64_t i, scail_bf16* hid, const float* m, const scail_dit_cond* cond,
                        const float* rope_cos, const float* rope_sin, int64_t Btot, int64_t Ltok, const BlockBufs& bf,
                        const scail_dit_sp* sp, int64_t row0, int64_t rows, bool pair, void* stream) {
    const scail_dit_config& c = h->cfg;
    const int64_t D = c.hidden_size, nh = c.num_heads, N = sp->ranks;
    const float eps = c.layernorm_epsilon;
    const int64_t Lf = N * Ltok, Lfp = (Lf + 63) / 64 * 64;
    const int64_t B = pair ? 1 : Btot;        // elements of the self-attention part
    scail_bf16 *qkv = bf.qkv, *q = qkv;
    const scail_dit_layer& lw = h->layers[i];
    DIT_TRY(scail_ln_modulate(hid, D, bf.xn, D, m, m + D, 6 * D, B, Ltok, Ltok, 0, D, eps, stream));
    if (sp->mode == SCAIL_SP_ULYSSES) {
        // head <-> sequence all-to-all of q, k, v after norm + RoPE (sat/mpu/ulysses_attn_layer.py:65-107), full-length attention on
        // heads / N heads, all-to-all back.  The CFG elements are independent sequences: element b + 1's projection runs under
        // element b's exchange, element b's attention under element b + 1's exchange, the ways back under the other's attention.
        const int64_t Hn = nh / N, Dn = Hn * 128, slab = Ltok * Dn;
        const bool side = sp->side_stream[0] != nullptr;
        auto st = [&](int64_t b) { return side ? sp->side_stream[b & 1] : stream; };
        SideJoin sj{h, sp, stream};
        if (side) {
            // one side stream per element (up to 4 ranks: an element's launches leave a partial last round of the chip which the other
            // element's kernels fill).  The collectives are still enqueued in the order fwd(0), fwd(1), back(0), back(1).
            for (hipEvent_t& e : h->sp_ev)
                if (e == nullptr && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                    scail_set_error("scail_dit (sequence parallel): hipEventCreate failed");
                    return 2;
                }
            if (hipEventRecord(h->sp_ev[0], (hipStream_t)stream) != hipSuccess) { scail_set_error("hipEventRecord failed"); return 2; }
            for (int j = 0; j < (B < 2 ? (int)B : 2); ++j) {
                if (hipStreamWaitEvent((hipStream_t)sp->side_stream[j], h->sp_ev[0], 0) != hipSuccess) { scail_set_error("hipStreamWaitEvent failed"); return 2; }
                sj.n = j + 1;
            }
        }
        for (int64_t b = 0; b < B; ++b) {
            void* s = st(b);
            scail_bf16* qb = qkv + b * Ltok * 3 * D;
            scail_bf16* send = sp->send + b * 3 * N * slab;             // [N][Ltok][3 Dn]: ONE message per destination rank, q | k | v side by side
            DIT_PROF_S(SCAIL_DIT_PROF_GEMM, s, scail_gemm_bf16(bf.xn + b * Ltok * D, D, lw.qkv_w, lw.qkv_b, qb, 3 * D, Ltok, 3 * D, D, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, s));
            // the norm + RoPE kernels write the send layout themselves; v is a plain copy into it
            DIT_TRY(scail_rmsnorm_rope_slabs(qb + D, 3 * D, send + Dn, Dn, 3 * Dn, 3 * slab, lw.kn, rope_cos, rope_sin, Ltok, Ltok, D, 128, eps, 1.0f, s));
            DIT_TRY(scail_rmsnorm_rope_slabs(qb, 3 * D, send, Dn, 3 * Dn, 3 * slab, lw.qn, rope_cos, rope_sin, Ltok, Ltok, D, 128, eps, ATTN_LOG2_SCALE, s));
            DIT_TRY(scail_rmsnorm_rope_slabs(qb + 2 * D, 3 * D, send + 2 * Dn, Dn, 3 * Dn, 3 * slab, nullptr, nullptr, nullptr, Ltok, Ltok, D, 128, eps, 1.0f, s));
            DIT_TRY(sp_exchange(sp, SCAIL_SP_FWD_START, i, b, s));
        }
        for (int64_t b = 0; b < B; ++b) {
            void* s = st(b);
            // exposed wait: the event pair brackets nothing but the stream's wait for the collective (0 when it finished under other work)
            DIT_PROF_S(SCAIL_DIT_PROF_XCH_FWD_WAIT, s, sp_exchange(sp, SCAIL_SP_FWD_WAIT, i, b, s));
            const scail_bf16* recv = sp->recv + b * 3 * N * slab;       // [N * Ltok][3 Dn]: all ranks' tokens (rank-major), q | k | v of my heads
            scail_bf16* vt = bf.vt + b * Hn * 128 * Lfp;
            scail_bf16* of = sp->ofull + b * N * slab;
            DIT_TRY(scail_transpose_v(recv + 2 * Dn, 3 * Dn, 0, vt, 1, Hn, 128, Lf, s));
            // two elements' launches run side by side on the two streams and fill each other's partial rounds: one 256-row launch each
            // (a lone launch gets the planned shape: whole 256-row rounds + 192-row tiles for the rest, csrc/attn.hip attn4_plan)
            struct Hint { Hint(int r) { scail_attn4_rows_hint(r); } ~Hint() { scail_attn4_rows_hint(0); } } hint(side && B > 1 ? 256 : 0);
            DIT_PROF_S(SCAIL_DIT_PROF_SELF_ATTN, s, scail_flash_attn_bf16(recv, 0, 3 * Dn, recv + Dn, 0, 0, 3 * Dn, vt, 0, 0, of, 0, Dn, 1, Hn, Lf, Lf, 1,
                                                                          SCAIL_ATTN_Q_PRESCALED, 0, s));
            DIT_TRY(sp_exchange(sp, SCAIL_SP_BACK_START, i, b, s));
        }
        for (int64_t b = 0; b < B; ++b) {
            void* s = st(b);
            DIT_PROF_S(SCAIL_DIT_PROF_XCH_BACK_WAIT, s, sp_exchange(sp, SCAIL_SP_BACK_WAIT, i, b, s));
            // back[b][g] = my tokens, head group g  ->  att rows (token, all columns)
            DIT_TRY(scail_slabs_to_rows(sp->back + b * N * slab, Dn, slab, bf.att + b * Ltok * D, D, Ltok, D, s));
        }
        if (sj.join() != 0) {
            scail_set_error("scail_dit (sequence parallel): joining the side streams failed");
            return 2;
        }
    } else {
        // ONE exchange per layer: all-gather of the post-norm, post-RoPE K rows and of the V rows; K / V projection per element first
        // so that element b's gather runs under element b + 1's projection and the Q projection of all elements.
        const int64_t rowsD = Ltok * D;
        for (int64_t b = 0; b < B; ++b) {
            scail_bf16* qb = qkv + b * Ltok * 3 * D;
            scail_bf16* send = sp->send + b * 2 * rowsD;                // [Ltok][2 D]: ONE message, k | v side by side
            const scail_bf16* xb = bf.xn + b * rowsD;
            DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(xb, D, lw.qkv_w + D * D, lw.qkv_b + D, qb + D, 3 * D, Ltok, D, D, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
            DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(xb, D, lw.qkv_w + 2 * D * D, lw.qkv_b + 2 * D, send + D, 2 * D, Ltok, D, D, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
            DIT_TRY(scail_rmsnorm_rope(qb + D, 3 * D, send, 2 * D, lw.kn, rope_cos, rope_sin, Ltok, Ltok, D, 128, eps, stream));
            DIT_TRY(sp_exchange(sp, SCAIL_SP_FWD_START, i, b, stream));
        }
        if (rows == Ltok) {
            DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(bf.xn, D, lw.qkv_w, lw.qkv_b, q, 3 * D, B * Ltok, D, D, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
            DIT_TRY(scail_rmsnorm_rope_scaled(q, 3 * D, q, 3 * D, lw.qn, rope_cos, rope_sin, B * Ltok, Ltok, D, 128, eps, ATTN_LOG2_SCALE, stream));
        } else {
            for (int64_t b = 0; b < B; ++b) {       // the wanted rows' queries only (their RoPE rows start at row0)
                scail_bf16* qr = q + (b * Ltok + row0) * 3 * D;
                DIT_PROF(SCAIL_DIT_PROF_GEMM, scail_gemm_bf16(bf.xn + (b * Ltok + row0) * D, D, lw.qkv_w, lw.qkv_b, qr, 3 * D, rows, D, D, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
                DIT_TRY(scail_rmsnorm_rope_scaled(qr, 3 * D, qr, 3 * D, lw.qn, rope_cos + row0 * 64, rope_sin + row0 * 64, rows, rows, D, 128, eps, ATTN_LOG2_SCALE, stream));
            }
        }
        for (int64_t b = 0; b < B; ++b) {
            DIT_PROF(SCAIL_DIT_PROF_XCH_FWD_WAIT, sp_exchange(sp, SCAIL_SP_FWD_WAIT, i, b, stream));
            const scail_bf16* recv = sp->recv + b * 2 * N * rowsD;      // [N * Ltok][2 D]: gathered rows (rank-major), k | v
            scail_bf16* vt = bf.vt + b * nh * 128 * Lfp;
            DIT_TRY(scail_transpose_v(recv + D, 2 * D, 0, vt, 1, nh, 128, Lf, stream));
            DIT_PROF(SCAIL_DIT_PROF_SELF_ATTN, scail_flash_attn_bf16(q + (b * Ltok + row0) * 3 * D, 0, 3 * D, recv, 0, 0, 2 * D, vt, 0, 0, bf.att + b * rowsD + row0 * D, 0, D,
                                                                     1, nh, rows, Lf, 1, SCAIL_ATTN_Q_PRESCALED, 0, stream));
        }
    }
    return block_tail(h, i, hid, m, cond, Btot, Ltok, bf.att, q, bf.xn, bf.ff, row0, rows, pair, stream);
}

// Seam B2 (SAT hook layer_forward): one block on caller-owned hidden states.  Workspace: scail_dit_block_workspace_bytes.
static int64_t block_ws(const scail_dit_config& c, int64_t B, int64_t Ltok, int sp_mode, int64_t ranks, int64_t* off) {
    const int64_t D = c.hidden_size, FF = c.inner_hidden_size;
    int64_t o = 0;
    const int64_t sizes[5] = {B * Ltok * D * 2, B * Ltok * 3 * D * 2, B * Ltok * D * 2, B * Ltok * FF * 2, vt_elems(c, B, Ltok, sp_mode, ranks) * 2};
    for (int j = 0; j < 5; ++j) { off[j] = o; o += align256(sizes[j]); }
    return o;
}
extern "C" int64_t scail_dit_block_workspace_bytes(const scail_dit* h, int64_t B, int64_t Ltok) {
    if (h == nullptr || B <= 0 || Ltok <= 0) return -1;
    int64_t off[5];
    return block_ws(h->cfg, B, Ltok, -1, 1, off);
}
extern "C" int64_t scail_dit_block_sp_workspace_bytes(const scail_dit* h, int32_t mode, int32_t ranks, int64_t B, int64_t Ltok) {
    if (h == nullptr || B <= 0 || Ltok <= 0 || ranks < 2 || (mode != SCAIL_SP_ALLGATHER && mode != SCAIL_SP_ULYSSES)) return -1;
    if (mode == SCAIL_SP_ULYSSES && h->cfg.num_heads % ranks != 0) return -1;
    int64_t off[5];
    return block_ws(h->cfg, B, Ltok, mode, ranks, off);
}
extern "C" int scail_dit_block_sp(scail_dit* h, int64_t layer, scail_bf16* hidden, const float* mod, const scail_dit_cond* cond,
                                  const float* rope_cos, const float* rope_sin, int64_t B, int64_t Ltok, const scail_dit_sp* sp,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
    SCAIL_REQUIRE(h != nullptr && hidden != nullptr && mod != nullptr && cond != nullptr, "null argument");
    SCAIL_REQUIRE(layer >= 0 && layer < h->cfg.num_layers && B > 0 && Ltok > 0, "bad layer / shape");
    DIT_TRY(sp_check(h, sp));
    int64_t off[5];
    const int64_t need = block_ws(h->cfg, B, Ltok, sp->mode, sp->ranks, off);
    SCAIL_REQUIRE(workspace != nullptr && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                  "workspace too small or not 256-byte aligned (scail_dit_block_sp_workspace_bytes)");
    char* base = static_cast<char*>(workspace);
    auto P = [&](int j) { return reinterpret_cast<scail_bf16*>(base + off[j]); };
    const BlockBufs bf{P(0), P(1), P(2), P(3), P(4)};
    return dit_block_sp(h, layer, hidden, mod, cond, rope_cos, rope_sin, B, Ltok, bf, sp, 0, Ltok, false, stream);
}
extern "C" int scail_dit_block(scail_dit* h, int64_t layer, scail_bf16* hidden, const float* mod, const scail_dit_cond* cond,
                               const float* rope_cos, const float* rope_sin, int64_t B, int64_t Ltok,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    SCAIL_REQUIRE(h != nullptr && hidden != nullptr && mod != nullptr && cond != nullptr, "null argument");
    SCAIL_REQUIRE(layer >= 0 && layer < h->cfg.num_layers && B > 0 && Ltok > 0, "bad layer / shape");
    int64_t off[5];
    const int64_t need = block_ws(h->cfg, B, Ltok, -1, 1, off);
    SCAIL_REQUIRE(workspace != nullptr && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                  "workspace too small or not 256-byte aligned (scail_dit_block_workspace_bytes)");
    char* base = static_cast<char*>(workspace);
    auto P = [&](int j) { return reinterpret_cast<scail_bf16*>(base + off[j]); };
    const BlockBufs bf{P(0), P(1), P(2), P(3), P(4)};
    return dit_block(h, layer, hidden, mod, cond, rope_cos, rope_sin, B, Ltok, bf, 0, Ltok, false, stream);
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
    if (h != nullptr) {
        for (ProfPool& p : h->pool)
            for (hipEvent_t e : p.ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : h->sp_ev)
            if (e != nullptr) (void)hipEventDestroy(e);
        if (h->restart_ctr != nullptr) (void)hipFree(h->restart_ctr);
    }
    delete h;
}

extern "C" int scail_dit_profile(scail_dit* h, int enable) {
    SCAIL_REQUIRE(h != nullptr, "null handle");
    h->prof = enable != 0;
    if (h->prof) {
        for (ProfPool& p : h->pool) p.used = 0;       // a new measurement: reuse the pooled events
        // restart counter of the self-attention (SCAIL_DIT_PROF_ATTN_RESTARTS): allocated once, zeroed per measurement (blocking calls:
        // profiling is enabled outside of stream captures)
        if (h->restart_ctr == nullptr && hipMalloc(reinterpret_cast<void**>(&h->restart_ctr), sizeof(uint32_t)) != hipSuccess) {
            h->restart_ctr = nullptr;
            scail_set_error("scail_dit_profile: hipMalloc of the restart counter failed");
            return 2;
        }
        if (hipMemset(h->restart_ctr, 0, sizeof(uint32_t)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            scail_set_error("scail_dit_profile: clearing the restart counter failed");
            return 2;
        }
    }
    return 0;
}

extern "C" int scail_dit_profile_read(scail_dit* h, int category, double* ms_total, int64_t* launches) {
    SCAIL_REQUIRE(h != nullptr && ms_total != nullptr && launches != nullptr, "null argument");
    if (category == SCAIL_DIT_PROF_ATTN_RESTARTS) {
        // workgroups of the timed self-attention launches that left the optimistic pass and ran again (scail_hip.h
        // scail_flash_attn_count_restarts); read after the launches have finished
        ProfPool& pa = h->pool[SCAIL_DIT_PROF_SELF_ATTN];
        uint32_t n = 0;
        if ((pa.used >= 2 && hipEventSynchronize(pa.ev[pa.used - 1]) != hipSuccess) || h->restart_ctr == nullptr ||
            hipDeviceSynchronize() != hipSuccess || hipMemcpy(&n, h->restart_ctr, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) {
            scail_set_error("scail_dit_profile_read: reading the restart counter failed (was profiling enabled?)");
            return 2;
        }
        *ms_total = 0.0;
        *launches = (int64_t)n;
        return 0;
    }
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
extern "C" int64_t scail_dit_sp_workspace_bytes(const scail_dit* h, int32_t mode, int32_t ranks, int64_t B, int64_t T, int64_t H, int64_t W) {
    if (h == nullptr || B <= 0 || T <= 0 || H <= 0 || W <= 0 || H % 4 != 0 || W % 4 != 0 || ranks < 2) return -1;
    if (mode != SCAIL_SP_ALLGATHER && mode != SCAIL_SP_ULYSSES) return -1;
    if (mode == SCAIL_SP_ULYSSES && h->cfg.num_heads % ranks != 0) return -1;
    return layout(h->cfg, B, T, H, W, mode, ranks).total;
}

static int dit_step_impl(scail_dit* h, const float* x, const float* timesteps, const scail_dit_cond* cond,
                         const scail_bf16* ref, int64_t n_ref, const scail_bf16* pose, int64_t n_pose,
                         const float* rope_cos, const float* rope_sin, float* out,
                         int64_t B, int64_t T, int64_t H, int64_t W, const scail_dit_sp* sp, uint32_t flags, void* workspace, int64_t workspace_bytes,
                         void* stream);

extern "C" int scail_dit_step(scail_dit* h, const float* x, const float* timesteps, const scail_dit_cond* cond,
                              const scail_bf16* ref, int64_t n_ref, const scail_bf16* pose, int64_t n_pose,
                              const float* rope_cos, const float* rope_sin, float* out,
                              int64_t B, int64_t T, int64_t H, int64_t W, uint32_t flags, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    return dit_step_impl(h, x, timesteps, cond, ref, n_ref, pose, n_pose, rope_cos, rope_sin, out, B, T, H, W, nullptr, flags, workspace, workspace_bytes, stream);
}

extern "C" int scail_dit_step_sp(scail_dit* h, const float* x, const float* timesteps, const scail_dit_cond* cond,
                                 const scail_bf16* ref, int64_t n_ref, const scail_bf16* pose, int64_t n_pose,
                                 const float* rope_cos, const float* rope_sin, float* out,
                                 int64_t B, int64_t T, int64_t H, int64_t W, const scail_dit_sp* sp, uint32_t flags, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
    SCAIL_REQUIRE(h != nullptr, "null handle");
    DIT_TRY(sp_check(h, sp));
    return dit_step_impl(h, x, timesteps, cond, ref, n_ref, pose, n_pose, rope_cos, rope_sin, out, B, T, H, W, sp, flags, workspace, workspace_bytes, stream);
}

// x (B, T, 16, H, W): the whole latent (sp == nullptr) or this rank's H- or W-slab of it (rope tables rank-shifted by the host)
static int dit_step_impl(scail_dit* h, const float* x, const float* timesteps, const scail_dit_cond* cond,
                         const scail_bf16* ref, int64_t n_ref, const scail_bf16* pose, int64_t n_pose,
                         const float* rope_cos, const float* rope_sin, float* out,
                         int64_t B, int64_t T, int64_t H, int64_t W, const scail_dit_sp* sp, uint32_t flags, void* workspace, int64_t workspace_bytes,
                         void* stream) {
    SCAIL_REQUIRE(h != nullptr && cond != nullptr, "null handle / conditioning");
    SCAIL_REQUIRE((flags & ~(uint32_t)SCAIL_DIT_CFG_PAIR) == 0, "unknown step flag");
    SCAIL_REQUIRE(!(flags & SCAIL_DIT_CFG_PAIR) || (B == 2 && n_ref == 1 && n_pose == 1),
                  "SCAIL_DIT_CFG_PAIR needs B == 2 with one shared ref / pose (element 1 = element 0 except for the conditioning)");
    SCAIL_REQUIRE(B > 0 && B <= 8 && T > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "latent batch must be 1..8, H and W multiples of 4");
    SCAIL_REQUIRE((n_ref == 1 || n_ref == B) && (n_pose == 1 || n_pose == B), "ref / pose batch must be 1 or B");
    SCAIL_REQUIRE(cond->Bc == 1 || cond->Bc == B, "clip batch must be 1 or B");
    const scail_dit_config& c = h->cfg;
    const Ws s = sp ? layout(c, B, T, H, W, sp->mode, sp->ranks) : layout(c, B, T, H, W);
    SCAIL_REQUIRE(c.time_embed_dim == c.hidden_size, "final-layer table add needs time_embed_dim == hidden_size");
    SCAIL_REQUIRE(workspace != nullptr && workspace_bytes >= s.total, "workspace too small (scail_dit_workspace_bytes)");
    SCAIL_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");

    const int64_t D = c.hidden_size, nl = c.num_layers;
    const float eps = c.layernorm_epsilon;
    const int64_t hp = H / 2, wp = W / 2;
    const int64_t Lref = hp * wp, Lnoise = T * hp * wp, Lpose = T * (H / 4) * (W / 4);
    const int64_t Ltok = Lref + Lnoise + Lpose, Lrn = Lref + Lnoise;
    char* base = static_cast<char*>(workspace);
    auto B16 = [&](int64_t off) { return reinterpret_cast<scail_bf16*>(base + off); };
    auto F32 = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
    scail_bf16 *tok = B16(s.tok), *hid = B16(s.h), *xn = B16(s.xn), *qkv = B16(s.qkv), *att = B16(s.att), *ff = B16(s.ff);
    scail_bf16 *vt = B16(s.vt), *xf = B16(s.xf), *tokout = B16(s.tokout);
    float *temb = F32(s.temb), *e1 = F32(s.e1), *emb = F32(s.emb), *adaln = F32(s.adaln), *mod = F32(s.mod);
    float *emb2 = F32(s.emb2), *fin = F32(s.fin);
    const scail_dit_weights& w = h->w;

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
    // SCAIL_DIT_CFG_PAIR: the two elements are one latent under two conditionings (VanillaCFG, guiders.py:41-57) and stay equal until the
    // first cross attention of layer 0 (dit...:1009-1042): patch embedding and layer 0 up to the self-attention residual run once
    const bool pair = (flags & SCAIL_DIT_CFG_PAIR) != 0 && nl > 1;
    const int64_t Be = pair ? 1 : B;
    DIT_TRY(scail_patchify(x, ref, pose, tok, Be, n_ref, n_pose, T, H, W, KPAD, stream));
    for (int64_t b = 0; b < Be; ++b) {
        DIT_TRY(scail_gemm_bf16(tok + b * Ltok * KPAD, KPAD, w.patch_w, w.patch_b, hid + b * Ltok * D, D, Lrn, D, KPAD,
                                SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
        DIT_TRY(scail_gemm_bf16(tok + (b * Ltok + Lrn) * KPAD, KPAD, w.pose_w, w.pose_b, hid + (b * Ltok + Lrn) * D, D, Lpose, D,
                                KPAD, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, stream));
    }

    const BlockBufs bf{xn, qkv, att, ff, vt};
    for (int64_t i = 0; i < nl; ++i) {
        // the last layer's output is only read at the noise tokens (final layer below): queries / out-projection / cross attention /
        // MLP of its ref and pose rows are skipped (23 % of that layer's post-K/V work; same result)
        const bool last = i == nl - 1;
        const int64_t row0 = last ? Lref : 0, rows = last ? Lnoise : Ltok;
        if (sp != nullptr) {
            DIT_TRY(dit_block_sp(h, i, hid, mod + i * B * 6 * D, cond, rope_cos, rope_sin, B, Ltok, bf, sp, row0, rows, pair && i == 0, stream));
        } else {
            DIT_TRY(dit_block(h, i, hid, mod + i * B * 6 * D, cond, rope_cos, rope_sin, B, Ltok, bf, row0, rows, pair && i == 0, stream));
        }
    }

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
        DIT_TRY(scail_dit_step(h, xin, timesteps + 2 * i, cond, ref, 1, pose, 1, rope_cos, rope_sin, v, 2, T, H, W, SCAIL_DIT_CFG_PAIR, workspace,
                               step_bytes, stream));
        DIT_TRY(scail_cfg_euler(x, v, n, cfg_scale, dsigma[i], stream));
    }
    return 0;
}

// Seam B4 in C (include/scail_vae.h): WanVAE_.encode / .decode (sgm/models/wan_vae.py:516-568) as a fixed sequence of
// operator launches -- the C++ statement of scail_amd/wan_vae.py.  Host code only; activations are channels-last
// (T, H, W, C) bf16 tensors living in NSLOT equally sized slots of the caller's workspace.
#include <cmath>
#include <vector>

#include "common.h"
#include "../../include/scail_vae.h"

struct scail_vae {
    scail_vae_weights w;
    std::vector<scail_vae_stage> enc, dec;
    scail_vae_trace_fn trace = nullptr;
    void* trace_user = nullptr;
};

namespace {

constexpr int NSLOT = 6;

inline int64_t align256(int64_t n) { return (n + 255) / 256 * 256; }

struct Tens {            // channels-last activation
    scail_bf16* p = nullptr;
    int64_t T = 0, H = 0, W = 0, C = 0;
    int slot = -1;
    int64_t vox() const { return T * H * W; }
};

struct Arena {
    char* base;
    int64_t slot_bytes;
    bool used[NSLOT] = {false, false, false, false, false, false};
    void* stream;
    int err = 0;
    scail_vae_trace_fn trace = nullptr;     // scail_vae_set_trace: called after every operator launch whose output is a whole activation
    void* trace_user = nullptr;
    int trace_n = 0;
    void emit(const char* op, const Tens& t) {
        if (trace) trace(trace_user, trace_n, op, t.p, t.T, t.H, t.W, t.C);
        ++trace_n;
    }
    Tens get(int64_t T, int64_t H, int64_t W, int64_t C) {
        Tens t;
        t.T = T; t.H = H; t.W = W; t.C = C;
        if (T * H * W * C * 2 > slot_bytes) { err = 1; scail_set_error("scail_vae: activation larger than a workspace slot"); return t; }
        for (int i = 0; i < NSLOT; ++i)
            if (!used[i]) { used[i] = true; t.slot = i; t.p = reinterpret_cast<scail_bf16*>(base + i * slot_bytes); return t; }
        err = 1;
        scail_set_error("scail_vae: out of workspace slots");
        return t;
    }
    void put(Tens& t) { if (t.slot >= 0) used[t.slot] = false; t.slot = -1; }
};

#define VAE_TRY(call_)                  \
    {                                   \
        const int rc_ = (call_);        \
        if (rc_ != 0) return rc_;       \
    }
#define VAE_CHK(a_) if ((a_).err) return 1;

// conv3d_cl with the defaults of scail_amd.ops.conv3d_cl: causal "same" padding (kt-1, kh/2, kw/2), stride 1
int conv(Arena& a, const Tens& x, const scail_conv_w& cw, Tens& out, int64_t To, int64_t Ho, int64_t Wo,
         int st = 1, int sh = 1, int sw = 1, int pt = -1, int ph = -1, int pw = -1, int ups = 0, int ot_mul = 1, int ot_off = 0,
         const Tens* resid = nullptr, bool alloc = true) {
    if (pt < 0) { pt = cw.kt - 1; ph = cw.kh / 2; pw = cw.kw / 2; }
    if (alloc) { out = a.get(To, Ho, Wo, cw.N); VAE_CHK(a) }
    SCAIL_REQUIRE(x.C == cw.Cin, "scail_vae: channel mismatch between an activation and its convolution");
    int32_t geom[21] = {(int32_t)x.T, (int32_t)x.H, (int32_t)x.W, (int32_t)x.C, (int32_t)To, (int32_t)Ho, (int32_t)Wo,
                        cw.kt, cw.kh, cw.kw, st, sh, sw, pt, ph, pw, ups, ot_mul, ot_off, cw.N, cw.Kpad};
    VAE_TRY(scail_conv3d_cl(x.p, cw.w, cw.b, out.p, out.C, resid ? resid->p : nullptr, resid ? resid->C : 0, geom, a.stream));
    if (ot_mul == 1) a.emit("conv", out);         // (the two interleaved halves of an upsample3d time_conv are seen through the resample conv)
    return 0;
}

// ResidualBlock (wan_vae.py:180-218); consumes x.
// xn (optional), on entry: SiLU(RMS_norm(x) * r.gamma0) if x's producer already wrote it (xn->p != nullptr; consumed), else empty.
// next_gamma: the RMS_norm weights of whatever reads this block's output next (the next ResidualBlock's residual.0, the decoder head's norm) or nullptr.
// Where ONE generated kernel covers "last convolution + shortcut sum + that consumer's norm" (scail_conv3d_kernel_for(.., 2): 96 channels at full
// resolution) the block's last call is scail_conv3d_cl_resid_norm: *xn returns the consumer's normalised input, and with need_raw = false (nobody reads
// the raw sum) x comes back EMPTY.  Everywhere else the sequence is what it was and *xn comes back empty.  scail_amd/wan_vae.py _res follows the same rule.
int res_block(Arena& a, const scail_vae_res& r, Tens& x, Tens* xn = nullptr, const float* next_gamma = nullptr, bool need_raw = true) {
    Tens h = x, y, y2, out;
    const bool sc = r.shortcut.w != nullptr;
    if (sc) VAE_TRY(conv(a, x, r.shortcut, h, x.T, x.H, x.W));
    if (xn != nullptr && xn->p != nullptr) {
        y = *xn;
        *xn = Tens();
    } else {
        y = a.get(x.T, x.H, x.W, x.C); VAE_CHK(a)
        VAE_TRY(scail_rms_silu(x.p, y.p, r.gamma0, x.vox(), x.C, 1, a.stream));
        a.emit("rms_silu", y);
    }
    int32_t geom[21] = {(int32_t)y.T, (int32_t)y.H, (int32_t)y.W, (int32_t)y.C, (int32_t)x.T, (int32_t)x.H, (int32_t)x.W,
                        3, 3, 3, 1, 1, 1, 2, 1, 1, 0, 1, 0, r.conv2.N, r.conv2.Kpad};
    if (r.conv2.kt == 3 && r.conv2.kh == 3 && r.conv2.kw == 3 && y.C % 32 == 0 && r.conv2.N <= 96 &&
        (scail_conv3d_kernel_for(geom, r.conv2.N, 0, 1) == 4 || scail_conv3d_kernel_for(geom, r.conv2.N, 0, 0) != 4)) {
        // conv -> RMS_norm -> SiLU in one kernel: the raw conv output never goes to HBM (same rule as ops.conv_norm_fusable): the generated
        // kernel's norm epilogue where it applies (N = 96: scail_conv4f_e4), else the hipcc halo kernel's -- except where the plain generated
        // kernel runs but its norm epilogue does not: there conv + a separate rms_silu pass beats the fused hipcc kernel (13.9 + 2.8 vs 21.0 ms
        // on the 96-channel full-resolution shape, round 3).
        y2 = a.get(x.T, x.H, x.W, r.conv2.N); VAE_CHK(a)
        VAE_TRY(scail_conv3d_cl_norm(y.p, r.conv2.w, r.conv2.b, y2.p, y2.C, r.gamma3, geom, a.stream));
        a.emit("conv_norm", y2);
        a.put(y);
    } else {
        VAE_TRY(conv(a, y, r.conv2, y2, x.T, x.H, x.W));
        a.put(y);
        VAE_TRY(scail_rms_silu(y2.p, y2.p, r.gamma3, y2.vox(), y2.C, 1, a.stream));
        a.emit("rms_silu", y2);
    }
    int32_t geom6[21] = {(int32_t)y2.T, (int32_t)y2.H, (int32_t)y2.W, (int32_t)y2.C, (int32_t)x.T, (int32_t)x.H, (int32_t)x.W,
                         r.conv6.kt, r.conv6.kh, r.conv6.kw, 1, 1, 1, r.conv6.kt - 1, r.conv6.kh / 2, r.conv6.kw / 2, 0, 1, 0, r.conv6.N, r.conv6.Kpad};
    if (xn != nullptr && next_gamma != nullptr && y2.C == r.conv6.Cin && scail_conv3d_kernel_for(geom6, r.conv6.N, h.C, 2) != 0) {
        Tens nrm = a.get(x.T, x.H, x.W, r.conv6.N); VAE_CHK(a)
        if (need_raw) { out = a.get(x.T, x.H, x.W, r.conv6.N); VAE_CHK(a) }
        VAE_TRY(scail_conv3d_cl_resid_norm(y2.p, r.conv6.w, r.conv6.b, need_raw ? out.p : nullptr, nrm.p, r.conv6.N, h.p, h.C, next_gamma, geom6, a.stream));
        if (need_raw) a.emit("conv", out);
        a.emit("conv_resid_norm", nrm);
        *xn = nrm;
    } else {
        VAE_TRY(conv(a, y2, r.conv6, out, x.T, x.H, x.W, 1, 1, 1, -1, -1, -1, 0, 1, 0, &h));
    }
    a.put(y2);
    if (sc) a.put(h);
    a.put(x);
    x = out;
    return 0;
}

// the norm weights the consumer of stage i's output applies first, when that consumer is a ResidualBlock of the table (else nullptr)
const float* next_res_gamma(const std::vector<scail_vae_stage>& st, size_t i) {
    return (i + 1 < st.size() && st[i + 1].kind == 0) ? st[i + 1].res.gamma0 : nullptr;
}

// AttentionBlock (wan_vae.py:221-262): per frame, single head over the H*W tokens; consumes x
int attn_block(Arena& a, const scail_vae_attn& at, Tens& x) {
    const int64_t T = x.T, nt = x.H * x.W, C = x.C, npad = (nt + 63) / 64 * 64, nt8 = (nt + 7) / 8 * 8;
    SCAIL_REQUIRE(C == at.C && nt <= 32768, "scail_vae: mid-block attention handles up to 32768 tokens per frame ((H/8)*(W/8))");
    Tens y = a.get(x.T, x.H, x.W, C); VAE_CHK(a)
    VAE_TRY(scail_rms_silu(x.p, y.p, at.gamma, x.vox(), C, 0, a.stream));
    a.emit("rms_silu", y);
    Tens tmp = a.get(1, 1, 1, 0); VAE_CHK(a)          // one slot, carved up below
    const int64_t act = align256((T * nt + 8) * C * 2);       // + 8 rows: the score GEMM reads k rows up to ceil8(nt) of the last frame
    SCAIL_REQUIRE(4 * act + align256(nt * npad * 2) + align256(C * npad * 2) <= a.slot_bytes, "scail_vae: attention temporaries exceed a slot");
    char* tb = reinterpret_cast<char*>(tmp.p);
    scail_bf16 *q = reinterpret_cast<scail_bf16*>(tb), *k = reinterpret_cast<scail_bf16*>(tb + act),
               *v = reinterpret_cast<scail_bf16*>(tb + 2 * act), *o = reinterpret_cast<scail_bf16*>(tb + 3 * act),
               *S = reinterpret_cast<scail_bf16*>(tb + 4 * act), *vt = reinterpret_cast<scail_bf16*>(tb + 4 * act + align256(nt * npad * 2));
    const int64_t M = T * nt;
    VAE_TRY(scail_gemm_bf16(y.p, C, at.q_w, at.q_b, q, C, M, C, C, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, a.stream));
    VAE_TRY(scail_gemm_bf16(y.p, C, at.k_w, at.k_b, k, C, M, C, C, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, a.stream));
    VAE_TRY(scail_gemm_bf16(y.p, C, at.v_w, at.v_b, v, C, M, C, C, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, a.stream));
    if (hipMemsetAsync(S, 0, nt * npad * 2, (hipStream_t)a.stream) != hipSuccess ||
        hipMemsetAsync(vt, 0, C * npad * 2, (hipStream_t)a.stream) != hipSuccess) {
        scail_set_error("scail_vae: hipMemsetAsync failed");
        return 2;
    }
    // in double, then rounded once: what Python's 1.0 / math.sqrt(C) hands the operator seam (and SDPA's default scale in the reference); the
    // float expression 1.0f / sqrtf(C) is one ulp off at C = 384, which made this executor differ from the layer path in the last bit
    const float scale = (float)(1.0 / std::sqrt((double)C));
    for (int64_t f = 0; f < T; ++f) {
        const scail_bf16 *qf = q + f * nt * C, *kf = k + f * nt * C, *vf = v + f * nt * C;
        // N = ceil8(nt): the up to 7 extra score columns come from the next frame's keys (or the pad rows) and are zeroed by the softmax
        VAE_TRY(scail_gemm_bf16(qf, C, kf, nullptr, S, npad, nt, nt8, C, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, a.stream));
        VAE_TRY(scail_softmax_rows(S, npad, nt, nt, scale, a.stream));
        VAE_TRY(scail_transpose2d(vf, C, nt * C, vt, npad, C * npad, nt, C, 1, a.stream));
        VAE_TRY(scail_gemm_bf16(S, npad, vt, nullptr, o + f * nt * C, C, nt, C, npad, SCAIL_EPI_BIAS, nullptr, 0, nullptr, 0, 0, a.stream));
    }
    Tens out = a.get(x.T, x.H, x.W, C); VAE_CHK(a)
    VAE_TRY(scail_gemm_bf16(o, C, at.proj_w, at.proj_b, out.p, C, M, C, C, SCAIL_EPI_RESID, x.p, C, nullptr, 0, 0, a.stream));
    a.emit("attn", out);
    a.put(tmp); a.put(y); a.put(x);
    x = out;
    return 0;
}

int copy_frame(Arena& a, scail_bf16* dst, const scail_bf16* src, int64_t elems) {
    if (hipMemcpyAsync(dst, src, elems * 2, hipMemcpyDeviceToDevice, (hipStream_t)a.stream) != hipSuccess) {
        scail_set_error("scail_vae: hipMemcpyAsync failed");
        return 2;
    }
    return 0;
}

// Resample downsample2d / downsample3d (wan_vae.py:87-96, :133-150); consumes x
int down_stage(Arena& a, const scail_vae_stage& s, Tens& x) {
    Tens y;
    VAE_TRY(conv(a, x, s.resample, y, x.T, x.H / 2, x.W / 2, 1, 2, 2, 0, 0, 0));
    a.put(x);
    if (s.temporal && y.T > 1) {
        const int64_t To = (y.T - 1) / 2;
        Tens out = a.get(1 + To, y.H, y.W, y.C); VAE_CHK(a)
        VAE_TRY(copy_frame(a, out.p, y.p, y.H * y.W * y.C));           // first frame bypasses the temporal conv (:146-148)
        VAE_TRY(conv(a, y, s.time_conv, out, To, y.H, y.W, 2, 1, 1, 0, 0, 0, 0, 1, 1, nullptr, false));
        a.put(y);
        y = out;
    }
    x = y;
    return 0;
}

// Resample upsample2d / upsample3d (wan_vae.py:76-85, :100-131); consumes x.  next_gamma / xn as in res_block: where ONE generated kernel covers the
// resample convolution + the next ResidualBlock's RMS_norm + SiLU, *xn returns that block's normalised input beside the raw output
int up_stage(Arena& a, const scail_vae_stage& s, Tens& x, Tens* xn = nullptr, const float* next_gamma = nullptr) {
    if (s.temporal && x.T > 1) {
        Tens t2 = a.get(1 + 2 * (x.T - 1), x.H, x.W, x.C); VAE_CHK(a)
        const int64_t fr = x.H * x.W * x.C;
        VAE_TRY(copy_frame(a, t2.p, x.p, fr));                             // 'Rep': the first latent frame is not doubled (:106-108)
        Tens tail = x;                                                     // frames >= 1 never see frame 0 (:120-130)
        tail.p = x.p + fr; tail.T = x.T - 1;
        VAE_TRY(conv(a, tail, s.time_conv0, t2, x.T - 1, x.H, x.W, 1, 1, 1, -1, -1, -1, 0, 2, 1, nullptr, false));
        VAE_TRY(conv(a, tail, s.time_conv1, t2, x.T - 1, x.H, x.W, 1, 1, 1, -1, -1, -1, 0, 2, 2, nullptr, false));
        a.put(x);
        x = t2;
    }
    Tens out;
    int32_t geom[21] = {(int32_t)x.T, (int32_t)x.H, (int32_t)x.W, (int32_t)x.C, (int32_t)x.T, (int32_t)(2 * x.H), (int32_t)(2 * x.W),
                        s.resample.kt, s.resample.kh, s.resample.kw, 1, 1, 1, 0, 1, 1, 1, 1, 0, s.resample.N, s.resample.Kpad};
    if (xn != nullptr && next_gamma != nullptr && x.C == s.resample.Cin && scail_conv3d_kernel_for(geom, s.resample.N, 0, 2) != 0) {
        out = a.get(x.T, 2 * x.H, 2 * x.W, s.resample.N); VAE_CHK(a)
        Tens nrm = a.get(x.T, 2 * x.H, 2 * x.W, s.resample.N); VAE_CHK(a)
        VAE_TRY(scail_conv3d_cl_resid_norm(x.p, s.resample.w, s.resample.b, out.p, nrm.p, s.resample.N, nullptr, 0, next_gamma, geom, a.stream));
        a.emit("conv", out);
        a.emit("conv_resid_norm", nrm);
        *xn = nrm;
    } else {
        VAE_TRY(conv(a, x, s.resample, out, x.T, 2 * x.H, 2 * x.W, 1, 1, 1, 0, 1, 1, 1));
    }
    a.put(x);
    x = out;
    return 0;
}

// a slot holds the largest activation (full resolution, `dim` channels) or the mid-block attention's temporaries
// (q, k, v, o of all latent frames + one frame's score matrix and V^T) -- the latter dominates for single-frame clips
int64_t slot_bytes_for(const scail_vae_weights& w, int64_t T, int64_t H, int64_t W) {
    const int64_t dim = w.enc_conv1.N > w.dec_head.Cin ? w.enc_conv1.N : w.dec_head.Cin;
    const int64_t Tl = 1 + (T - 1) / 4, nt = (H / 8) * (W / 8), npad = (nt + 63) / 64 * 64;
    const int64_t C = w.enc_attn.C > w.dec_attn.C ? w.enc_attn.C : w.dec_attn.C;
    const int64_t attn = 4 * align256((Tl * nt + 8) * C * 2) + align256(nt * npad * 2) + align256(C * npad * 2);
    const int64_t act = align256(T * H * W * dim * 2);
    return act > attn ? act : attn;
}

}  // namespace

extern "C" int scail_vae_create(const scail_vae_weights* w, scail_vae** out) {
    SCAIL_REQUIRE(w != nullptr && out != nullptr, "null argument");
    SCAIL_REQUIRE(w->n_enc >= 0 && w->n_dec >= 0 && (w->n_enc == 0 || w->enc != nullptr) && (w->n_dec == 0 || w->dec != nullptr), "bad stage tables");
    SCAIL_REQUIRE(w->z_dim > 0 && w->z_dim % 8 == 0, "z_dim must be a positive multiple of 8");
    scail_vae* h = new scail_vae;
    h->w = *w;
    h->enc.assign(w->enc, w->enc + w->n_enc);
    h->dec.assign(w->dec, w->dec + w->n_dec);
    h->w.enc = h->enc.data();
    h->w.dec = h->dec.data();
    *out = h;
    return 0;
}

extern "C" void scail_vae_destroy(scail_vae* h) { delete h; }

extern "C" int scail_vae_set_trace(scail_vae* h, scail_vae_trace_fn fn, void* user) {
    SCAIL_REQUIRE(h != nullptr, "null handle");
    h->trace = fn;
    h->trace_user = user;
    return 0;
}

extern "C" int64_t scail_vae_workspace_bytes(const scail_vae* h, int64_t T, int64_t H, int64_t W) {
    if (h == nullptr || T <= 0 || (T - 1) % 4 != 0 || H <= 0 || W <= 0 || H % 8 != 0 || W % 8 != 0) return -1;
    return NSLOT * slot_bytes_for(h->w, T, H, W);
}

extern "C" int scail_vae_encode(scail_vae* h, const float* video, float* latent, int64_t T, int64_t H, int64_t W,
                                void* workspace, int64_t workspace_bytes, void* stream) {
    SCAIL_REQUIRE(h != nullptr && video != nullptr && latent != nullptr, "null argument");
    SCAIL_REQUIRE(T > 0 && (T - 1) % 4 == 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "video needs T = 1 + 4n frames and H, W multiples of 8");
    const scail_vae_weights& w = h->w;
    Arena a;
    a.base = static_cast<char*>(workspace);
    a.slot_bytes = slot_bytes_for(w, T, H, W);
    a.stream = stream;
    a.trace = h->trace; a.trace_user = h->trace_user;
    SCAIL_REQUIRE(workspace != nullptr && workspace_bytes >= NSLOT * a.slot_bytes && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                  "workspace too small or not 256-byte aligned (scail_vae_workspace_bytes)");
    Tens x = a.get(T, H, W, 8); VAE_CHK(a)
    VAE_TRY(scail_to_channels_last(video, x.p, nullptr, nullptr, 3, 8, T * H * W, stream));
    Tens y;
    Tens xn;       // the next ResidualBlock's normalised input, when its producer wrote it (res_block, here the stem)
    {
        const scail_conv_w& cw = w.enc_conv1;
        int32_t geom[21] = {(int32_t)T, (int32_t)H, (int32_t)W, (int32_t)x.C, (int32_t)T, (int32_t)H, (int32_t)W,
                            cw.kt, cw.kh, cw.kw, 1, 1, 1, cw.kt - 1, cw.kh / 2, cw.kw / 2, 0, 1, 0, cw.N, cw.Kpad};
        const float* g0 = (!h->enc.empty() && h->enc[0].kind == 0) ? h->enc[0].res.gamma0 : nullptr;
        if (g0 != nullptr && x.C == cw.Cin && scail_conv3d_kernel_for(geom, cw.N, 0, 2) != 0) {
            y = a.get(T, H, W, cw.N); VAE_CHK(a)
            xn = a.get(T, H, W, cw.N); VAE_CHK(a)
            VAE_TRY(scail_conv3d_cl_resid_norm(x.p, cw.w, cw.b, y.p, xn.p, cw.N, nullptr, 0, g0, geom, stream));
            a.emit("conv", y);
            a.emit("conv_resid_norm", xn);
        } else {
            VAE_TRY(conv(a, x, w.enc_conv1, y, T, H, W));
        }
    }
    a.put(x);
    x = y;
    for (size_t i = 0; i < h->enc.size(); ++i) {
        const scail_vae_stage& s = h->enc[i];
        if (s.kind == 0) { VAE_TRY(res_block(a, s.res, x, &xn, next_res_gamma(h->enc, i))); }
        else if (s.kind == 1) { VAE_TRY(down_stage(a, s, x)); }
        else { scail_set_error("scail_vae_encode: upsampling stage in the encoder table"); return 1; }
    }
    VAE_TRY(res_block(a, w.enc_mid0, x));
    VAE_TRY(attn_block(a, w.enc_attn, x));
    VAE_TRY(res_block(a, w.enc_mid2, x));
    VAE_TRY(scail_rms_silu(x.p, x.p, w.enc_head_gamma, x.vox(), x.C, 1, stream));
    a.emit("rms_silu", x);
    VAE_TRY(conv(a, x, w.enc_head, y, x.T, x.H, x.W));
    a.put(x);
    x = y;
    VAE_TRY(conv(a, x, w.conv1, y, x.T, x.H, x.W));
    a.put(x);
    x = y;
    SCAIL_REQUIRE(x.T == 1 + (T - 1) / 4 && x.H == H / 8 && x.W == W / 8, "scail_vae_encode: stage table does not compress (4, 8, 8)");
    return scail_from_channels_last(x.p, x.C, latent, w.enc_scale, w.enc_shift, w.z_dim, x.vox(), -3.0e38f, 3.0e38f, stream);
}

extern "C" int scail_vae_decode(scail_vae* h, const float* latent, float* video, int64_t Tl, int64_t hl, int64_t wl,
                                void* workspace, int64_t workspace_bytes, void* stream) {
    SCAIL_REQUIRE(h != nullptr && video != nullptr && latent != nullptr, "null argument");
    SCAIL_REQUIRE(Tl > 0 && hl > 0 && wl > 0, "bad latent shape");
    const scail_vae_weights& w = h->w;
    const int64_t T = 1 + 4 * (Tl - 1), H = 8 * hl, W = 8 * wl;
    Arena a;
    a.base = static_cast<char*>(workspace);
    a.slot_bytes = slot_bytes_for(w, T, H, W);
    a.stream = stream;
    a.trace = h->trace; a.trace_user = h->trace_user;
    SCAIL_REQUIRE(workspace != nullptr && workspace_bytes >= NSLOT * a.slot_bytes && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                  "workspace too small or not 256-byte aligned (scail_vae_workspace_bytes)");
    Tens x = a.get(Tl, hl, wl, w.z_dim); VAE_CHK(a)
    VAE_TRY(scail_to_channels_last(latent, x.p, w.dec_scale, w.dec_shift, w.z_dim, w.z_dim, Tl * hl * wl, stream));
    Tens y;
    VAE_TRY(conv(a, x, w.conv2, y, x.T, x.H, x.W));
    a.put(x);
    x = y;
    VAE_TRY(conv(a, x, w.dec_conv1, y, x.T, x.H, x.W));
    a.put(x);
    x = y;
    VAE_TRY(res_block(a, w.dec_mid0, x));
    VAE_TRY(attn_block(a, w.dec_attn, x));
    VAE_TRY(res_block(a, w.dec_mid2, x));
    Tens xn;       // the next consumer's normalised input, when its producer wrote it (res_block)
    for (size_t i = 0; i < h->dec.size(); ++i) {
        const scail_vae_stage& s = h->dec[i];
        const bool last = i + 1 == h->dec.size();      // the last block feeds the head's norm and nothing else
        if (s.kind == 0) { VAE_TRY(res_block(a, s.res, x, &xn, last ? w.dec_head_gamma : next_res_gamma(h->dec, i), !last)); }
        else if (s.kind == 2) { VAE_TRY(up_stage(a, s, x, &xn, next_res_gamma(h->dec, i))); }
        else { scail_set_error("scail_vae_decode: downsampling stage in the decoder table"); return 1; }
    }
    if (xn.p != nullptr) {             // the head's RMS_norm + SiLU came out of the last block's epilogue
        if (x.p != nullptr) a.put(x);
        x = xn;
    } else {
        VAE_TRY(scail_rms_silu(x.p, x.p, w.dec_head_gamma, x.vox(), x.C, 1, stream));
        a.emit("rms_silu", x);
    }
    VAE_TRY(conv(a, x, w.dec_head, y, x.T, x.H, x.W));
    a.put(x);
    x = y;
    SCAIL_REQUIRE(x.T == T && x.H == H && x.W == W, "scail_vae_decode: stage table does not expand (4, 8, 8)");
    return scail_from_channels_last(x.p, x.C, video, nullptr, nullptr, 3, x.vox(), -3.0e38f, 3.0e38f, stream);
}

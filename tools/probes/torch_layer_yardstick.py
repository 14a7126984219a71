"""Yardstick only (never on the product path, imports nothing from this repo): ONE transformer block of the config-2 step
written the way the reference runs it -- plain PyTorch ops on the GPU (F.linear -> hipBLASLt, F.scaled_dot_product_attention,
F.layer_norm, separate elementwise modulate / gate / residual / RoPE passes; dit_video_crossattn_sc_xc.py:1009-1203,
sat/transformer_defaults.py:47-79) -- timed on the same box.  x 40 layers = what the reference's own computation costs
per sampler step on an MI355X with the library kernels PyTorch ships."""
import json
import torch
import torch.nn.functional as F

dev = "cuda"
B, L, D, H, FF, Lt, Lc = 2, 48832, 5120, 40, 13824, 512, 257
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.02).to(bf)
W = dict(qkv=r(3 * D, D), o=r(D, D), cq=r(D, D), ckv=r(2 * D, D), clipkv=r(2 * D, D), co=r(D, D), w1=r(FF, D), w2=r(D, FF))
bias = {k: torch.zeros(v.shape[0], device=dev, dtype=bf) for k, v in W.items()}
nw = {k: torch.ones(D, device=dev, dtype=bf) for k in ("q", "k", "cq", "ck", "clipk")}
ln_w, ln_b = torch.ones(D, device=dev, dtype=bf), torch.zeros(D, device=dev, dtype=bf)
h = torch.randn(B, L, D, device=dev, generator=g).to(bf)
text, clip = r(B, Lt, D) * 50, r(B, Lc, D) * 50
mod = [torch.randn(B, 1, D, device=dev, generator=g).to(bf) * 0.1 for _ in range(6)]
cos, sin = torch.rand(L, 128, device=dev, generator=g).to(bf), torch.rand(L, 128, device=dev, generator=g).to(bf)


def rms(x, w):                      # RMSNorm over the full hidden dim, fp32 inside (dit...:48-68)
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(bf) * w


def heads(x):
    return x.view(x.shape[0], x.shape[1], H, 128).transpose(1, 2)


def rope(x):                        # interleaved pairs, bf16 tables (dit...:336-340, 553-557)
    x1, x2 = x[..., 0::2], x[..., 1::2]
    rot = torch.stack((-x2, x1), dim=-1).flatten(-2)
    return x * cos + rot * sin


def block(h):
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod
    x = F.layer_norm(h, (D,), eps=1e-6) * (1 + sc_a) + sh_a
    q, k, v = F.linear(x, W["qkv"], bias["qkv"]).chunk(3, -1)
    q, k = rope(heads(rms(q, nw["q"]))), rope(heads(rms(k, nw["k"])))
    a = F.scaled_dot_product_attention(q, k, heads(v)).transpose(1, 2).reshape(B, L, D)
    h = h + g_a * F.linear(a, W["o"], bias["o"])
    x = F.layer_norm(h, (D,), ln_w, ln_b, eps=1e-6)
    q = heads(rms(F.linear(x, W["cq"], bias["cq"]), nw["cq"]))
    kt, vt = F.linear(text, W["ckv"], bias["ckv"]).chunk(2, -1)
    kc, vc = F.linear(clip, W["clipkv"], bias["clipkv"]).chunk(2, -1)
    a = F.scaled_dot_product_attention(q, heads(rms(kt, nw["ck"])), heads(vt)) + \
        F.scaled_dot_product_attention(q, heads(rms(kc, nw["clipk"])), heads(vc))
    h = h + F.linear(a.transpose(1, 2).reshape(B, L, D), W["co"], bias["co"])
    x = F.layer_norm(h, (D,), eps=1e-6) * (1 + sc_m) + sh_m
    return h + g_m * F.linear(F.gelu(F.linear(x, W["w1"], bias["w1"]), approximate="tanh"), W["w2"], bias["w2"])


with torch.no_grad():
    for _ in range(2):
        o = block(h)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); o = block(h); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
ms = sorted(ts)[1]
print(json.dumps(dict(case="one config-2 transformer block in plain PyTorch-ROCm ops (reference-style)", ms_per_layer=ms,
                      s_per_step_40_layers=ms * 40 / 1e3, latent_tokens_per_s=37632 / (ms * 40 / 1e3), finite=bool(torch.isfinite(o.float()).all()))))

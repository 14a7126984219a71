"""CPU oracle for the SCAIL video-DiT sampling hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the product path
(``scail_amd``) never does and fails loudly when its HIP library is missing.

This is a plain fp32 (torch-on-CPU, no autocast) restatement of the reference's
algorithm for the path named by BASELINE.json / SURVEY.md section 8.  Every
function cites the reference file:line (paths relative to /root/reference) it
follows.  Nothing here imports the reference; ``oracle/gen_golden.py`` runs
the *real* reference in the build container and stores its outputs under
``tests/golden/`` -- ``tests/test_oracle_golden.py`` pins this restatement to
those vectors.

Parity status: PINNED against outputs of the reference itself (the reference
ships no tests/golden vectors of its own -- SURVEY.md section 4).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------
@dataclass
class DiTConfig:
    """Mirror of the ``network_config.params`` block of
    configs/video_model/Wan2.1-i2v-14Bsc-pose-xc-latent.yaml:21-81."""

    hidden_size: int = 5120
    num_layers: int = 40
    num_attention_heads: int = 40
    inner_hidden_size: int = 13824
    text_dim: int = 4096
    time_freq_dim: int = 256
    time_embed_dim: int = 5120
    in_channels: int = 20
    out_channels: int = 16
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    clip_dim: int = 1280
    layernorm_epsilon: float = 1e-6
    # Rotary3DPositionEmbeddingMixin: height/width = latent_{height,width}//patch
    latent_height: int = 300
    latent_width: int = 300
    num_frames: int = 81
    time_compressed_rate: int = 4
    theta: float = 10000.0
    global_rope_H: int = 0     # dit_video_crossattn_sc_xc.py:1570
    global_rope_W: int = 120   # dit_video_crossattn_sc_xc.py:1571

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def rope_grid(self) -> Tuple[int, int, int]:
        """(T, H, W) extents of the rotary tables (dit...:1390-1393, 424-426)."""
        t = (self.num_frames - 1) // self.time_compressed_rate + 1
        return t, self.latent_height // self.patch_size[1], self.latent_width // self.patch_size[2]


# Test-size configs.  head_dim is 128 in both (the shipped 14B and 1.3B configs both use 128 and
# the HIP attention kernel is specialised for it).  TINY has two heads so head indexing is
# exercised; CONFIG1 is BASELINE.json configs[0] ("2-layer/128-dim ... 4x8x8 latent").
TINY = dict(hidden_size=256, num_layers=2, num_attention_heads=2, inner_hidden_size=512,
            text_dim=64, time_freq_dim=256, time_embed_dim=256, latent_height=32, latent_width=32,
            num_frames=13)
CONFIG1 = dict(hidden_size=128, num_layers=2, num_attention_heads=1, inner_hidden_size=256,
               text_dim=64, time_freq_dim=256, time_embed_dim=128, latent_height=32, latent_width=32,
               num_frames=13)


def state_dict_spec(cfg: DiTConfig) -> Dict[str, Tuple[int, ...]]:
    """Names/shapes of the reference state_dict (verified by
    loading into the reference with strict=True in oracle/gen_golden.py)."""
    D, FF, Dt = cfg.hidden_size, cfg.inner_hidden_size, cfg.time_embed_dim
    pk = cfg.in_channels
    p = cfg.patch_size
    out = {}
    out["mixins.patch_embed.proj.weight"] = (D, pk, *p)
    out["mixins.patch_embed.proj.bias"] = (D,)
    out["mixins.patch_embed.proj_pose.weight"] = (D, pk, *p)
    out["mixins.patch_embed.proj_pose.bias"] = (D,)
    for i in range(cfg.num_layers):
        out[f"mixins.adaln_layer.adaLN_modulations.{i}"] = (1, 6, D)
    for nm in ("query", "key", "cross_query", "cross_key", "clip_feature_key"):
        for i in range(cfg.num_layers):
            out[f"mixins.adaln_layer.{nm}_layernorm_list.{i}.weight"] = (D,)
    for i in range(cfg.num_layers):
        out[f"mixins.adaln_layer.clip_feature_key_value_list.{i}.weight"] = (2 * D, D)
        out[f"mixins.adaln_layer.clip_feature_key_value_list.{i}.bias"] = (2 * D,)
    out["mixins.final_layer.adaLN_modulation"] = (1, 2, D)
    out["mixins.final_layer.linear.weight"] = (p[0] * p[1] * p[2] * cfg.out_channels, D)
    out["mixins.final_layer.linear.bias"] = (p[0] * p[1] * p[2] * cfg.out_channels,)
    for i in range(cfg.num_layers):
        L = f"transformer.layers.{i}."
        out[L + "attention.query_key_value.weight"] = (3 * D, D)
        out[L + "attention.query_key_value.bias"] = (3 * D,)
        out[L + "attention.dense.weight"] = (D, D)
        out[L + "attention.dense.bias"] = (D,)
        out[L + "cross_attention.query.weight"] = (D, D)
        out[L + "cross_attention.query.bias"] = (D,)
        out[L + "cross_attention.key_value.weight"] = (2 * D, D)
        out[L + "cross_attention.key_value.bias"] = (2 * D,)
        out[L + "cross_attention.dense.weight"] = (D, D)
        out[L + "cross_attention.dense.bias"] = (D,)
        out[L + "post_cross_attention_layernorm.weight"] = (D,)
        out[L + "post_cross_attention_layernorm.bias"] = (D,)
        out[L + "mlp.dense_h_to_4h.weight"] = (FF, D)
        out[L + "mlp.dense_h_to_4h.bias"] = (FF,)
        out[L + "mlp.dense_4h_to_h.weight"] = (D, FF)
        out[L + "mlp.dense_4h_to_h.bias"] = (D,)
    out["time_embed.0.weight"] = (Dt, cfg.time_freq_dim)
    out["time_embed.0.bias"] = (Dt,)
    out["time_embed.2.weight"] = (Dt, Dt)
    out["time_embed.2.bias"] = (Dt,)
    out["adaln_projection.1.weight"] = (6 * D, Dt)
    out["adaln_projection.1.bias"] = (6 * D,)
    out["text_embedding.0.weight"] = (D, cfg.text_dim)
    out["text_embedding.0.bias"] = (D,)
    out["text_embedding.2.weight"] = (D, D)
    out["text_embedding.2.bias"] = (D,)
    C = cfg.clip_dim
    out["clip_proj.proj.0.weight"] = (C,)
    out["clip_proj.proj.0.bias"] = (C,)
    out["clip_proj.proj.1.weight"] = (C, C)
    out["clip_proj.proj.1.bias"] = (C,)
    out["clip_proj.proj.3.weight"] = (D, C)
    out["clip_proj.proj.3.bias"] = (D,)
    out["clip_proj.proj.4.weight"] = (D,)
    out["clip_proj.proj.4.bias"] = (D,)
    return out


def make_state_dict(cfg: DiTConfig, seed: int = 1234, bf16_exact: bool = True) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights (no checkpoint is available offline).

    Distributions follow the reference init (sat/mpu/utils.py:89-94 N(0,0.02) for
    linears; dit...:888-893, 814-816 randn/sqrt(D) for the AdaLN tables) except that
    biases and norm weights are perturbed too, so that a kernel which drops a bias or
    an affine weight is caught.  Values are rounded to bf16-representable numbers so the
    fp32 oracle and the bf16 HIP path see the *same* weights.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    D = cfg.hidden_size
    for name, shape in state_dict_spec(cfg).items():
        if "adaLN_modulation" in name:
            w = torch.randn(shape, generator=g) / math.sqrt(D)
        elif ("layernorm" in name and name.endswith("weight")) \
                or name in ("clip_proj.proj.0.weight", "clip_proj.proj.4.weight"):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            w = 0.02 * torch.randn(shape, generator=g)
        elif "patch_embed" in name or "final_layer.linear" in name:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            w = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        else:
            # linears: N(0, 0.02) is the reference init; scale a little with fan-in so
            # tiny configs are not numerically degenerate
            fan_in = shape[-1]
            w = torch.randn(shape, generator=g) * max(0.02, 0.5 / math.sqrt(fan_in))
        if bf16_exact:
            w = w.to(torch.bfloat16).to(torch.float32)
        sd[name] = w
    return sd


# ----------------------------------------------------------------------------
# elementary ops
# ----------------------------------------------------------------------------
def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """sgm/modules/diffusionmodules/util.py:207-231 -- fp64 frequencies, fp32 t,
    product/cos/sin evaluated in fp64, [cos | sin] order."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float64) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb.to(torch.float32)


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU(approximate='tanh') (dit...:1296, 1339)."""
    return F.gelu(x, approximate="tanh")


def layer_norm(x: torch.Tensor, eps: float, w: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None):
    """sat/ops/layernorm.py:16-24 (torch LayerNorm over the last dim)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """dit...:48-68 RMSNorm over the FULL hidden dim (hidden_size_head: 5120, yaml :72)."""
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def modulate(x, shift, scale):
    """dit...:760-761."""
    return x * (1 + scale) + shift


# ----------------------------------------------------------------------------
# 3-segment 3D RoPE tables (dit...:382-513, 525-645)
# ----------------------------------------------------------------------------
def rope_dims(head_dim: int) -> Tuple[int, int, int]:
    """dit...:404-406."""
    dim_t = head_dim - 4 * (head_dim // 6)
    dim_h = (head_dim // 6) * 2
    dim_w = (head_dim // 6) * 2
    return dim_t, dim_h, dim_w


def _axis_freqs(dim: int, theta: float) -> torch.Tensor:
    return 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))


def rope_angles(cfg: DiTConfig, t_pos: torch.Tensor, h_pos: torch.Tensor, w_pos: torch.Tensor) -> torch.Tensor:
    """Angle tensor (T,H,W,head_dim) for explicit integer positions; interleaved layout
    ``repeat(.., '... n -> ... (n r)', r=2)`` (dit...:448-459)."""
    dt, dh, dw = rope_dims(cfg.head_dim)
    ft = torch.einsum("p,f->pf", t_pos.float(), _axis_freqs(dt, cfg.theta)).repeat_interleave(2, dim=-1)
    fh = torch.einsum("p,f->pf", h_pos.float(), _axis_freqs(dh, cfg.theta)).repeat_interleave(2, dim=-1)
    fw = torch.einsum("p,f->pf", w_pos.float(), _axis_freqs(dw, cfg.theta)).repeat_interleave(2, dim=-1)
    T, H, W = ft.shape[0], fh.shape[0], fw.shape[0]
    return torch.cat([
        ft[:, None, None, :].expand(T, H, W, -1),
        fh[None, :, None, :].expand(T, H, W, -1),
        fw[None, None, :, :].expand(T, H, W, -1),
    ], dim=-1)


def rope_tables(cfg: DiTConfig, rope_T: int, rope_H: int, rope_W: int,
                H_shift: int = 0, W_shift: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin tables (L, head_dim) for the token order [ref | noise | pose].

    * noise: t = 1..rope_T (grid_t, dit...:424), h/w = shift + 0..  (reshape_freq :543-551)
    * ref  : t = 0 (grid_extended_t, :428), same h/w window (:579-588)
    * pose : window h in [gH+Hs, gH+Hs+rope_H), w in [gW+Ws, gW+Ws+rope_W) of the
      *noise* tables, then ``avg_pool2d(2)`` of cos and of sin separately (:616-637).
    """
    hp = torch.arange(H_shift, H_shift + rope_H)
    wp = torch.arange(W_shift, W_shift + rope_W)
    ang_noise = rope_angles(cfg, torch.arange(1, rope_T + 1), hp, wp)
    ang_ref = rope_angles(cfg, torch.tensor([0]), hp, wp)
    hp2 = torch.arange(cfg.global_rope_H + H_shift, cfg.global_rope_H + H_shift + rope_H)
    wp2 = torch.arange(cfg.global_rope_W + W_shift, cfg.global_rope_W + W_shift + rope_W)
    ang_pose = rope_angles(cfg, torch.arange(1, rope_T + 1), hp2, wp2)

    def pool(x):  # (T,H,W,D) -> (T,H/2,W/2,D)
        return F.avg_pool2d(x.permute(0, 3, 1, 2), kernel_size=2, stride=2).permute(0, 2, 3, 1)

    hd = cfg.head_dim
    cos = torch.cat([ang_ref.cos().reshape(-1, hd), ang_noise.cos().reshape(-1, hd),
                     pool(ang_pose.cos()).reshape(-1, hd)], dim=0)
    sin = torch.cat([ang_ref.sin().reshape(-1, hd), ang_noise.sin().reshape(-1, hd),
                     pool(ang_pose.sin()).reshape(-1, hd)], dim=0)
    return cos.contiguous(), sin.contiguous()


def rope_tables_multi(cfg: DiTConfig, rope_T: int, rope_H: int, rope_W: int, n_char: int,
                      H_shift: int = 0, W_shift: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """EXTENSION, not in the reference (BASELINE config 5, SURVEY 8d: the reference has exactly one reference frame and one
    pose stream, dit...:1559): token order [ref_0 .. ref_{C-1} | noise | pose_0 .. pose_{C-1}].  Character 0 uses the
    reference's positions; character k > 0 takes windows the reference leaves unused:
      ref_k : t = 0, w window shifted by k * global_rope_W          (ref_0 at 0, the t = 0 plane is otherwise empty)
      pose_k: t = 1..T, w window shifted by global_rope_W + k * rope_W (pose_0 at global_rope_W, pose_k right of it)
    Parity for n_char > 1 is therefore unpinned by construction; n_char == 1 equals rope_tables()."""
    hd = cfg.head_dim
    hp = torch.arange(H_shift, H_shift + rope_H)
    wp = torch.arange(W_shift, W_shift + rope_W)

    def pool(x):
        return F.avg_pool2d(x.permute(0, 3, 1, 2), kernel_size=2, stride=2).permute(0, 2, 3, 1)

    cos, sin = [], []
    for k in range(n_char):
        a = rope_angles(cfg, torch.tensor([0]), hp, wp + k * cfg.global_rope_W)
        cos.append(a.cos().reshape(-1, hd)); sin.append(a.sin().reshape(-1, hd))
    a = rope_angles(cfg, torch.arange(1, rope_T + 1), hp, wp)
    cos.append(a.cos().reshape(-1, hd)); sin.append(a.sin().reshape(-1, hd))
    for k in range(n_char):
        a = rope_angles(cfg, torch.arange(1, rope_T + 1), hp + cfg.global_rope_H, wp + cfg.global_rope_W + k * rope_W)
        cos.append(pool(a.cos()).reshape(-1, hd)); sin.append(pool(a.sin()).reshape(-1, hd))
    return torch.cat(cos, 0).contiguous(), torch.cat(sin, 0).contiguous()


def rotate_half_interleaved(x: torch.Tensor) -> torch.Tensor:
    """dit...:336-340: (x0,x1,x2,x3,..) -> (-x1,x0,-x3,x2,..)."""
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


def apply_rope(t: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """dit...:556-557. t: (B, heads, L, hd); cos/sin: (L, hd)."""
    return t * cos[None, None] + rotate_half_interleaved(t) * sin[None, None]


# ----------------------------------------------------------------------------
# network
# ----------------------------------------------------------------------------
def sdpa(q, k, v):
    """sat/transformer_defaults.py:47-79 -> F.scaled_dot_product_attention, no mask,
    scale 1/sqrt(head_dim).  Written out so the oracle does not depend on a backend."""
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    return torch.matmul(torch.softmax(s, dim=-1), v)


def _heads(x: torch.Tensor, n: int) -> torch.Tensor:
    B, L, D = x.shape
    return x.view(B, L, n, D // n).permute(0, 2, 1, 3)


def _merge(x: torch.Tensor) -> torch.Tensor:
    B, n, L, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, L, n * d)


def patch_embed(cfg, sd, x20, ref20, pose20):
    """dit...:99-130.  Inputs (B,T,20,H,W); Conv3d k=s=(1,2,2) == per-patch linear with
    k index (c, p, q); token order (t h w)."""

    def proj(z, w, b):
        B, T, C, H, W = z.shape
        p, q = cfg.patch_size[1], cfg.patch_size[2]
        z = z.reshape(B, T, C, H // p, p, W // q, q).permute(0, 1, 3, 5, 2, 4, 6)
        z = z.reshape(B, T * (H // p) * (W // q), C * p * q)
        return z @ w.reshape(w.shape[0], -1).t() + b

    emb = proj(torch.cat([ref20, x20], dim=1), sd["mixins.patch_embed.proj.weight"], sd["mixins.patch_embed.proj.bias"])
    pemb = proj(pose20, sd["mixins.patch_embed.proj_pose.weight"], sd["mixins.patch_embed.proj_pose.bias"])
    return torch.cat([emb, pemb], dim=1)


def clip_proj(cfg, sd, clip):
    """MLPProj dit...:31-45: LN -> Linear -> GELU(erf) -> Linear -> LN (eps 1e-5 defaults)."""
    x = F.layer_norm(clip, (clip.shape[-1],), sd["clip_proj.proj.0.weight"], sd["clip_proj.proj.0.bias"], 1e-5)
    x = F.linear(x, sd["clip_proj.proj.1.weight"], sd["clip_proj.proj.1.bias"])
    x = F.gelu(x)
    x = F.linear(x, sd["clip_proj.proj.3.weight"], sd["clip_proj.proj.3.bias"])
    return F.layer_norm(x, (x.shape[-1],), sd["clip_proj.proj.4.weight"], sd["clip_proj.proj.4.bias"], 1e-5)


def text_embedding(cfg, sd, ctx):
    """dit...:1337-1341, 1505."""
    x = F.linear(ctx, sd["text_embedding.0.weight"], sd["text_embedding.0.bias"])
    return F.linear(gelu_tanh(x), sd["text_embedding.2.weight"], sd["text_embedding.2.bias"])


def time_embeddings(cfg, sd, timesteps):
    """dit...:1521-1524, 1555: emb = time_embed(sincos(t)); adaln_emb = Linear(SiLU(emb))."""
    t_emb = timestep_embedding(timesteps, cfg.time_freq_dim)
    emb = F.linear(t_emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    adaln = F.linear(F.silu(emb), sd["adaln_projection.1.weight"], sd["adaln_projection.1.bias"])
    return emb, adaln


def self_attention(cfg, sd, i, x, cos, sin, kv_gather=None):
    """dit...:1058-1105 (+ rotary hook :653-757, SDPA transformer_defaults.py:47-79).

    ``kv_gather`` (optional) maps this shard's post-RoPE (k, v) to the full-sequence
    (k, v): the sequence-parallel exchange."""
    L = f"transformer.layers.{i}.attention."
    eps = cfg.layernorm_epsilon
    qkv = F.linear(x, sd[L + "query_key_value.weight"], sd[L + "query_key_value.bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = rms_norm(q, sd[f"mixins.adaln_layer.query_layernorm_list.{i}.weight"], eps)
    k = rms_norm(k, sd[f"mixins.adaln_layer.key_layernorm_list.{i}.weight"], eps)
    n = cfg.num_attention_heads
    q, k, v = _heads(q, n), _heads(k, n), _heads(v, n)
    q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    if kv_gather is not None:
        k, v = kv_gather(k, v)
    ctx = _merge(sdpa(q, k, v))
    return F.linear(ctx, sd[L + "dense.weight"], sd[L + "dense.bias"])


def cross_attention(cfg, sd, i, x, text, clip):
    """dit...:1107-1203: q from x; (k,v) from text tokens and from CLIP tokens; RMSNorm on
    q, k, k_clip; two un-masked SDPAs summed; dense."""
    L = f"transformer.layers.{i}.cross_attention."
    eps = cfg.layernorm_epsilon
    n = cfg.num_attention_heads
    q = F.linear(x, sd[L + "query.weight"], sd[L + "query.bias"])
    k, v = F.linear(text, sd[L + "key_value.weight"], sd[L + "key_value.bias"]).chunk(2, dim=-1)
    kc, vc = F.linear(clip, sd[f"mixins.adaln_layer.clip_feature_key_value_list.{i}.weight"],
                      sd[f"mixins.adaln_layer.clip_feature_key_value_list.{i}.bias"]).chunk(2, dim=-1)
    q = rms_norm(q, sd[f"mixins.adaln_layer.cross_query_layernorm_list.{i}.weight"], eps)
    k = rms_norm(k, sd[f"mixins.adaln_layer.cross_key_layernorm_list.{i}.weight"], eps)
    kc = rms_norm(kc, sd[f"mixins.adaln_layer.clip_feature_key_layernorm_list.{i}.weight"], eps)
    q = _heads(q, n)
    ctx = _merge(sdpa(q, _heads(k, n), _heads(v, n))) + _merge(sdpa(q, _heads(kc, n), _heads(vc, n)))
    return F.linear(ctx, sd[L + "dense.weight"], sd[L + "dense.bias"])


def mlp(cfg, sd, i, x):
    """sat/transformer_defaults.py:163-176 non-gated branch; GELU-tanh (dit...:1295-1298)."""
    L = f"transformer.layers.{i}.mlp."
    h = gelu_tanh(F.linear(x, sd[L + "dense_h_to_4h.weight"], sd[L + "dense_h_to_4h.bias"]))
    return F.linear(h, sd[L + "dense_4h_to_h.weight"], sd[L + "dense_4h_to_h.bias"])


def block(cfg, sd, i, h, adaln_emb, text, clip, cos, sin, kv_gather=None):
    """AdaLNMixin.layer_forward dit...:1009-1051 (share_adaln branch :1025-1028)."""
    eps = cfg.layernorm_epsilon
    B, D = h.shape[0], cfg.hidden_size
    mod = adaln_emb.view(B, 6, D) + sd[f"mixins.adaln_layer.adaLN_modulations.{i}"]
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
    a_in = modulate(layer_norm(h, eps), sh_a, sc_a)
    h = h + g_a * self_attention(cfg, sd, i, a_in, cos, sin, kv_gather)
    Lp = f"transformer.layers.{i}.post_cross_attention_layernorm."
    c_in = layer_norm(h, eps, sd[Lp + "weight"], sd[Lp + "bias"])
    h = h + cross_attention(cfg, sd, i, c_in, text, clip)
    m_in = modulate(layer_norm(h, eps), sh_m, sc_m)
    h = h + g_m * mlp(cfg, sd, i, m_in)
    return h


def final_layer(cfg, sd, h, emb, ref_len, seq_len, rope_T, rope_H, rope_W):
    """FinalLayerMixin.final_forward + unpatchify dit...:818-835, 764-784."""
    shift, scale = (emb.unsqueeze(1) + sd["mixins.final_layer.adaLN_modulation"]).chunk(2, dim=1)
    x = modulate(layer_norm(h, cfg.layernorm_epsilon), shift, scale)
    x = F.linear(x, sd["mixins.final_layer.linear.weight"], sd["mixins.final_layer.linear.bias"])
    x = x[:, ref_len:ref_len + seq_len]
    B = x.shape[0]
    o, p, q = cfg.patch_size
    c = cfg.out_channels
    x = x.reshape(B, rope_T, rope_H, rope_W, o, p, q, c)
    # b (t h w) (o p q c) -> b (t o) c (h p) (w q)
    x = x.permute(0, 1, 4, 7, 2, 5, 3, 6).reshape(B, rope_T * o, c, rope_H * p, rope_W * q)
    return x


def dit_forward(cfg: DiTConfig, sd, x, timesteps, context, ref_concat, concat_smpl_render,
                image_clip_features, H_shift: int = 0, W_shift: int = 0, kv_gather=None,
                return_hidden: bool = False):
    """DiffusionTransformer.forward dit...:1452-1587 (the ``concat_images is not None``
    branch the CLI always takes, sample_video.py:455-463).

    x (B,T,16,H,W); timesteps (B,); context (B,Lt,text_dim); ref_concat (1|B,1,16,H,W);
    concat_smpl_render (1|B,T,16,H/2,W/2); image_clip_features (1|B,257,1280).
    EXTENSION (BASELINE config 5, not in the reference): ref_concat with C > 1 frames and concat_smpl_render with C*T
    frames are C characters, token order [ref_0..ref_{C-1} | noise | pose_0..pose_{C-1}], RoPE of rope_tables_multi().
    """
    B, T, C, H, W = x.shape
    x = x.float()

    def rep(z):
        return z.float().repeat(B // z.shape[0], *([1] * (z.dim() - 1))) if z.shape[0] != B else z.float()

    ref = rep(ref_concat)
    pose = rep(concat_smpl_render)
    x20 = torch.cat([x, torch.zeros(B, T, 4, H, W)], dim=2)                       # :1468,1503
    n_char = ref.shape[1]
    assert pose.shape[1] == n_char * T
    ref20 = torch.cat([ref, torch.ones(B, n_char, 4, H, W)], dim=2)               # :1483-1486
    pose20 = torch.cat([pose, torch.ones(B, n_char * T, 4, H // 2, W // 2)], dim=2)   # :1496-1501
    text = text_embedding(cfg, sd, context.float())
    clip = rep(clip_proj(cfg, sd, image_clip_features.float()))
    emb, adaln = time_embeddings(cfg, sd, timesteps)
    pt, ph, pw = cfg.patch_size
    rope_T, rope_H, rope_W = T // pt, H // ph, W // pw
    seq_len = T * H * W // (pt * ph * pw)
    ref_len = n_char * H * W // (pt * ph * pw)
    cos, sin = (rope_tables(cfg, rope_T, rope_H, rope_W, H_shift, W_shift) if n_char == 1 else
                rope_tables_multi(cfg, rope_T, rope_H, rope_W, n_char, H_shift, W_shift))
    h = patch_embed(cfg, sd, x20, ref20, pose20)
    hidden = [h]
    for i in range(cfg.num_layers):
        h = block(cfg, sd, i, h, adaln, text, clip, cos, sin, kv_gather)
        hidden.append(h)
    out = final_layer(cfg, sd, h, emb, ref_len, seq_len, rope_T, rope_H, rope_W)
    if return_hidden:
        return out, hidden
    return out


# ----------------------------------------------------------------------------
# sampler stack
# ----------------------------------------------------------------------------
def flow_sigmas(num_steps: int, shift_scale: float = 5.0) -> torch.Tensor:
    """make_flow_timesteps(0, n, shift_scale, mode='normal') sampling.py:888-903:
    s = linspace(0,1,n+1) (float64); s/(shift + s - shift*s) -> float32; sigma = 1 - that."""
    import numpy as np
    s = np.linspace(0.0, 1.0, num_steps + 1, endpoint=True)
    s = s / (shift_scale + s - shift_scale * s)
    return 1 - torch.tensor(s, dtype=torch.float32)


def cfg_combine(v_u, v_c, scale: float):
    """VanillaCFG.__call__ / NoDynamicThresholding guiders.py:41-45, sampling_utils.py:7-10."""
    return v_u + scale * (v_c - v_u)


def sample(cfg: DiTConfig, sd, x0, cond_ctx, uncond_ctx, ref_concat, concat_smpl_render,
           image_clip_features, num_steps=50, cfg_scale=4.0, shift_scale=5.0, network=None):
    """RFSampler.__call__ + denoise + sampler_step sampling.py:950-982 with Denoiser/RFScaling
    (denoiser.py:25-43, denoiser_scaling.py:71-78: c_in=1, c_out=1, c_skip=0, c_noise=1000*sigma)
    and VanillaCFG.prepare_inputs (guiders.py:47-57: batch = [uncond, cond]).

    x0: (1,T,16,H,W) fp32 noise.  Returns the final latent (fp32) and the per-step x list.
    ``network`` defaults to the oracle DiT; tests may pass the HIP network instead."""
    sig = flow_sigmas(num_steps, shift_scale)
    x = x0.clone().float()
    ctx = torch.cat([uncond_ctx, cond_ctx], dim=0)
    traj = []
    for i in range(num_steps):
        xin = torch.cat([x, x], dim=0)
        t = torch.stack([sig[i], sig[i]]) * 1000.0
        if network is None:
            v = dit_forward(cfg, sd, xin, t, ctx, ref_concat, concat_smpl_render, image_clip_features)
        else:
            v = network(xin, t, ctx)
        v = v.float()
        v_u, v_c = v.chunk(2)
        x = x + (sig[i + 1] - sig[i]) * cfg_combine(v_u, v_c, cfg_scale)
        traj.append(x.clone())
    return x, traj


def tile_weight(segment_length: int) -> torch.Tensor:
    """Triangular blending weight of a temporal tile, sampling.py:1038-1040."""
    w = (torch.arange(segment_length, dtype=torch.float32) + 0.5) * 2.0 / segment_length
    return torch.minimum(w, 2.0 - w)


def sample_long(cfg: DiTConfig, sd, x0, cond_ctx, uncond_ctx, ref_concat, smpl_tiled, image_clip_features,
                tile_indices, num_steps=50, cfg_scale=4.0, shift_scale=5.0, network=None):
    """RFSamplerLong.__call__ / sampler_step, sampling.py:1036-1085: every step denoises overlapping temporal
    tiles of the latent (frame index lists ``tile_indices``) against the matching pose tile ``smpl_tiled[:, k]``
    and blends the CFG-combined predictions with triangular weights before ONE Euler update of the whole latent.
    The reference walks the PAIRS (k, k+1), so interior tiles enter the weighted sum twice (and are denoised twice);
    the restatement keeps that multiplicity in the sums and evaluates each tile once.

    x0 (1,T,16,H,W) fp32; smpl_tiled (1, n_tiles, Ttile, 16, H/2, W/2).  Needs >= 2 tiles like the reference
    (one tile leaves weight_sum = 0 there)."""
    n = len(tile_indices)
    if n < 2:
        raise ValueError("RFSamplerLong needs at least two temporal tiles")
    sig = flow_sigmas(num_steps, shift_scale)
    x = x0.clone().float()
    ctx = torch.cat([uncond_ctx, cond_ctx], dim=0)
    w = tile_weight(len(tile_indices[0]))
    for i in range(num_steps):
        den = torch.zeros_like(x)
        wsum = torch.zeros(x.shape[1])
        t = torch.stack([sig[i], sig[i]]) * 1000.0
        for k in range(n):
            idx = list(tile_indices[k])
            mult = (1 if k == 0 else 2) if k < n - 1 else 1          # pairs (k-1,k) and (k,k+1)
            xin = torch.cat([x[:, idx], x[:, idx]], dim=0)
            pose = smpl_tiled[:, k]
            if network is None:
                v = dit_forward(cfg, sd, xin, t, ctx, ref_concat, pose, image_clip_features)
            else:
                v = network(xin, t, ctx, pose)
            v_u, v_c = v.float().chunk(2)
            d = cfg_combine(v_u, v_c, cfg_scale)
            den[:, idx] += mult * d * w[:, None, None, None]
            wsum[idx] += mult * w
        den = den / wsum[:, None, None, None]
        x = x + (sig[i + 1] - sig[i]) * den
    return x

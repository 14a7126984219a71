"""CPU oracles (fp32, plain torch) for the conditioning encoders -- TEST INFRASTRUCTURE ONLY.
Restate reference sgm/modules/encoders/umt5.py (T5Encoder) and clip.py (VisionTransformer, use_31_block);
pinned to outputs of the real reference classes by tests/golden/encoders_tiny.npz (oracle/gen_golden_encoders.py)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def t5_bucket(lq, lk, num_buckets=32, max_dist=128):
    """umt5.py:236-268 (bidirectional)."""
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    me = nb // 2
    large = me + (torch.log(rel.float() / me) / math.log(max_dist / me) * (nb - me)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < me, rel, large)


def t5_norm(x, w, eps=1e-6):
    """T5LayerNorm umt5.py:56-69."""
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def t5_gelu(x):
    """umt5.py:49-53."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def t5_encoder(sd, ids, mask, num_heads, num_layers, num_buckets=32):
    """T5Encoder.forward umt5.py:304-316 with T5SelfAttention (:162-176), T5Attention (:88-123: no score scaling,
    per-layer position bias, masked keys filled with finfo.min), T5FeedForward (:140-145)."""
    x = sd["token_embedding.weight"][ids]
    B, Ln, D = x.shape
    bucket = t5_bucket(Ln, Ln, num_buckets)
    for i in range(num_layers):
        p = f"blocks.{i}."
        h = t5_norm(x, sd[p + "norm1.weight"])
        q, k, v = (F.linear(h, sd[p + f"attn.{c}.weight"]).view(B, Ln, num_heads, -1) for c in "qkv")
        bias = sd[p + "pos_embedding.embedding.weight"][bucket].permute(2, 0, 1).unsqueeze(0).expand(B, -1, -1, -1).clone()
        bias.masked_fill_(mask.view(B, 1, 1, -1) == 0, torch.finfo(x.dtype).min)
        att = torch.softmax(torch.einsum("binc,bjnc->bnij", q, k) + bias, dim=-1)
        a = torch.einsum("bnij,bjnc->binc", att, v).reshape(B, Ln, -1)
        x = x + F.linear(a, sd[p + "attn.o.weight"])
        h = t5_norm(x, sd[p + "norm2.weight"])
        f = F.linear(h, sd[p + "ffn.fc1.weight"]) * t5_gelu(F.linear(h, sd[p + "ffn.gate.0.weight"]))
        x = x + F.linear(f, sd[p + "ffn.fc2.weight"])
    return t5_norm(x, sd["norm.weight"])


def clip_visual_31(sd, imgs, num_heads, num_layers, patch, eps=1e-5):
    """VisionTransformer.forward(use_31_block=True) clip.py:307-326 on already resized + normalised images
    (B,3,S,S): conv patch embedding (no bias, pre_norm), cls token, position embedding, pre-LN, first
    num_layers-1 pre-norm blocks (attention scale 1/sqrt(head_dim), GELU-erf MLP; :129-171, 71-108)."""
    x = F.conv2d(imgs, sd["patch_embedding.weight"], None, stride=patch).flatten(2).permute(0, 2, 1)
    B, _, D = x.shape
    x = torch.cat([sd["cls_embedding"].expand(B, -1, -1), x], dim=1) + sd["pos_embedding"]
    x = F.layer_norm(x, (D,), sd["pre_norm.weight"], sd["pre_norm.bias"], eps)
    hd = D // num_heads
    for i in range(num_layers - 1):
        p = f"transformer.{i}."
        h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        qkv = F.linear(h, sd[p + "attn.to_qkv.weight"], sd[p + "attn.to_qkv.bias"]).view(B, -1, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        a = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) / math.sqrt(hd), dim=-1) @ qkv[2]
        x = x + F.linear(a.permute(0, 2, 1, 3).reshape(B, -1, D), sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
        x = x + F.linear(h, sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
    return x


def clip_preprocess(frames):
    """CLIPModel.visual preprocessing clip.py:511-522: frames (B,3,H,W) in [-1,1] -> bicubic 224 -> [0,1] -> normalise."""
    x = F.interpolate(frames.float(), size=(224, 224), mode="bicubic", align_corners=False)
    x = x * 0.5 + 0.5
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    return (x - mean) / std

"""UMT5-XXL text encoder on MI355X (reference sgm/modules/encoders/umt5.py).

``T5Encoder`` mirrors the reference class (:270-316): same constructor arguments and state_dict keys
(``token_embedding.weight``, ``blocks.N.{norm1,attn.{q,k,v,o},norm2,ffn.{gate.0,fc1,fc2},pos_embedding.embedding}``,
``norm.weight``), so ``models_t5_umt5-xxl-enc-bf16.pth`` loads unchanged.  ``T5EncoderModel`` mirrors the
conditioner wrapper (:475-535); the tokenizer (scail_amd/tokenizer.py) needs files that are not available offline, so it is
only built when ``tokenizer_path`` is given (``encode_text(texts)``); ``forward`` takes token ids + mask.  Runs once per prompt (24 layers x 512 tokens = 4.8 TFLOP); all GEMMs go through
``scail_gemm_bf16`` (q,k,v fused into one GEMM; gate GEMM with the tanh-GELU epilogue; residual adds fused
into the o / fc2 GEMMs), T5LayerNorm through ``scail_rmsnorm_rope`` (same formula, eps 1e-6), attention with
the per-layer relative-position bias and key mask through ``scail_attn_small``."""
from __future__ import annotations

import math
from typing import Dict

import torch
from torch import nn

from . import lib as L
from . import ops
from .dit import _register


def relative_position_bucket(lq: int, lk: int, num_buckets: int = 32, max_dist: int = 128) -> torch.Tensor:
    """Bidirectional T5 buckets, umt5.py:236-268 (int32 (lq, lk))."""
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    out += torch.where(rel < max_exact, rel, large)
    return out.to(torch.int32)


class T5Encoder(nn.Module):
    def __init__(self, vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, shared_pos=True, dropout=0.1,
                 device=None, init_seed=7):
        super().__init__()
        if shared_pos:
            raise NotImplementedError("umt5_xxl uses shared_pos=False (umt5.py:468)")
        if dim % 64 or dim_attn % 64 or dim_ffn % 64 or (dim_attn // num_heads) % 8:
            raise NotImplementedError("dims must be multiples of 64, head_dim of 8")
        self.dim, self.dim_attn, self.dim_ffn = dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets = num_heads, num_layers, num_buckets
        dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        g = torch.Generator(device=dev).manual_seed(init_seed)
        rn = lambda shape, std: torch.randn(shape, device=dev, generator=g) * std
        spec = {"token_embedding.weight": ((vocab, dim), 1.0), "norm.weight": ((dim,), None)}
        for i in range(num_layers):
            p = f"blocks.{i}."
            spec[p + "norm1.weight"] = ((dim,), None)
            spec[p + "attn.q.weight"] = ((dim_attn, dim), (dim * dim_attn) ** -0.5)
            spec[p + "attn.k.weight"] = ((dim_attn, dim), dim ** -0.5)
            spec[p + "attn.v.weight"] = ((dim_attn, dim), dim ** -0.5)
            spec[p + "attn.o.weight"] = ((dim, dim_attn), dim_attn ** -0.5)
            spec[p + "norm2.weight"] = ((dim,), None)
            spec[p + "ffn.gate.0.weight"] = ((dim_ffn, dim), dim ** -0.5)
            spec[p + "ffn.fc1.weight"] = ((dim_ffn, dim), dim ** -0.5)
            spec[p + "ffn.fc2.weight"] = ((dim, dim_ffn), dim_ffn ** -0.5)
            spec[p + "pos_embedding.embedding.weight"] = ((num_buckets, num_heads), (2 * num_buckets * num_heads) ** -0.5)
        for n, (shape, std) in spec.items():
            w = torch.ones(shape, device=dev) if std is None else rn(shape, std)
            _register(self, n, nn.Parameter(w.to(torch.bfloat16), requires_grad=False))
        self._prepared = None
        self._buckets: Dict = {}

    def load_state_dict(self, *a, **k):
        self._prepared = None
        return super().load_state_dict(*a, **k)

    def prepare(self):
        if self._prepared is None:
            sd = {k: v.detach() for k, v in self.named_parameters()}
            if sd["norm.weight"].device.type != "cuda":
                raise L.ScailHipError("scail_amd.umt5 must live on the GPU (no CPU path)")
            L.load()
            layers = []
            for i in range(self.num_layers):
                p = f"blocks.{i}."
                layers.append(dict(
                    n1=sd[p + "norm1.weight"].float().contiguous(), n2=sd[p + "norm2.weight"].float().contiguous(),
                    qkv=torch.cat([sd[p + f"attn.{c}.weight"] for c in "qkv"], 0).to(torch.bfloat16).contiguous(),
                    o=sd[p + "attn.o.weight"].to(torch.bfloat16).contiguous(),
                    gate=sd[p + "ffn.gate.0.weight"].to(torch.bfloat16).contiguous(),
                    fc1=sd[p + "ffn.fc1.weight"].to(torch.bfloat16).contiguous(),
                    fc2=sd[p + "ffn.fc2.weight"].to(torch.bfloat16).contiguous(),
                    pos=sd[p + "pos_embedding.embedding.weight"].float().contiguous()))
            self._prepared = dict(layers=layers, norm=sd["norm.weight"].float().contiguous(), emb=sd["token_embedding.weight"])
        return self._prepared

    @torch.no_grad()
    def forward(self, ids: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
        """ids (B, L) int64, mask (B, L) {0,1} -> (B, L, dim) bf16 (umt5.py:304-316; dropout is identity in eval)."""
        W = self.prepare()
        B, Ln = ids.shape
        dev = W["emb"].device
        x = W["emb"].index_select(0, ids.reshape(-1).to(dev)).to(torch.bfloat16).view(B, Ln, self.dim).contiguous()
        if Ln not in self._buckets:
            self._buckets[Ln] = relative_position_bucket(Ln, Ln, self.num_buckets).to(dev).contiguous()
        bucket = self._buckets[Ln]
        km = mask.to(dev).to(torch.int32).contiguous() if mask is not None else None
        A = self.dim_attn
        for lw in W["layers"]:
            xn = ops.rmsnorm_rope(x, lw["n1"], out=torch.empty_like(x), eps=1e-6)
            qkv = ops.gemm(xn, lw["qkv"])
            a = ops.attn_small(qkv[..., :A], qkv[..., A:2 * A], qkv[..., 2 * A:], self.num_heads, scale=1.0, bucket=bucket,
                               bias_tab=lw["pos"], key_mask=km)
            ops.gemm(a, lw["o"], out=x, epilogue=L.EPI_RESID, resid=x)
            xn = ops.rmsnorm_rope(x, lw["n2"], out=xn, eps=1e-6)
            gate = ops.gemm(xn, lw["gate"], epilogue=L.EPI_GELU_TANH)
            h = ops.gemm(xn, lw["fc1"])
            ops.mul_(h, gate)
            ops.gemm(h, lw["fc2"], out=x, epilogue=L.EPI_RESID, resid=x)
        return ops.rmsnorm_rope(x, W["norm"], out=torch.empty_like(x), eps=1e-6)


def umt5_xxl_encoder(device="cuda", **kw) -> T5Encoder:
    """umt5.py:459-472 (encoder half)."""
    cfg = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32, shared_pos=False)
    cfg.update(kw)
    return T5Encoder(device=device, **cfg)


class T5EncoderModel(nn.Module):
    """Conditioner-side wrapper (umt5.py:475-535) without the tokenizer: ``forward(ids, mask)`` returns the encoder
    states with padded rows zeroed (``context * mask[:, :, None]``, :522)."""

    def __init__(self, max_length=512, checkpoint_path=None, device="cuda", tokenizer_path=None, dtype=None,
                 varlen_text=False, uncond_text_length=1, cond_length_multiple=1, **encoder_kwargs):
        super().__init__()
        self.max_length = max_length
        self.varlen_text, self.uncond_text_length, self.cond_length_multiple = varlen_text, uncond_text_length, cond_length_multiple
        self.model = umt5_xxl_encoder(device=device, **encoder_kwargs).eval()
        if checkpoint_path is not None and __import__("os").path.exists(checkpoint_path):
            self.model.load_state_dict(torch.load(checkpoint_path, map_location="cpu"))
        self.tokenizer = None
        if tokenizer_path is not None:          # umt5.py:509-510; files are not in this image, so only when given
            from .tokenizer import HuggingfaceTokenizer
            self.tokenizer = HuggingfaceTokenizer(name=tokenizer_path, seq_len=max_length, clean="whitespace")

    @torch.no_grad()
    def forward(self, ids, mask=None):
        """(ids, mask) token tensors, or -- like the reference's ``__call__(texts)`` -- a string / list of strings."""
        if mask is None and (isinstance(ids, str) or (isinstance(ids, (list, tuple)) and all(isinstance(u, str) for u in ids))):
            return self.encode_text(ids)
        ctx = self.model(ids, mask)
        return ops.row_affine(ctx, rowscale=mask.to(ctx.device).float().reshape(-1).contiguous())

    @torch.no_grad()
    def encode_text(self, texts):
        """The reference's ``__call__(texts)`` (umt5.py:512-535): tokenise, encode, zero the padded rows, and with
        ``varlen_text`` cut a single prompt to its own length (rounded up to ``cond_length_multiple``; the empty prompt to
        ``uncond_text_length``)."""
        if self.tokenizer is None:
            raise RuntimeError("T5EncoderModel was built without tokenizer_path; pass token ids to forward(ids, mask)")
        ids, mask = self.tokenizer(texts, return_mask=True, add_special_tokens=True)
        dev = next(self.model.parameters()).device
        z = self.forward(ids.to(dev), mask.to(dev))
        if self.varlen_text:
            if z.shape[0] != 1:
                raise AssertionError("varlen_text handles one prompt at a time")         # :524
            n = int(mask[0].sum())
            pad = (-n) % self.cond_length_multiple if n > 1 else self.uncond_text_length - n
            z = z[:, :n + pad]
        return z

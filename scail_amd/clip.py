"""CLIP ViT-H/14 visual tower on MI355X (reference sgm/modules/encoders/clip.py).

``VisionTransformer`` mirrors the reference class (:237-328; state_dict keys ``patch_embedding.weight``,
``cls_embedding``, ``pos_embedding``, ``pre_norm.*``, ``transformer.N.{norm1,attn.to_qkv,attn.proj,norm2,mlp.0,mlp.2}``,
``post_norm.*``, ``head``) and implements the only branch SCAIL uses: ``forward(x, use_31_block=True)`` -- the
257 x 1280 tokens after the first 31 blocks (:323-325).  ``CLIPModel.visual(videos)`` mirrors :511-526
(bicubic resize to 224 and normalisation on the host side like the reference, then the tower).
Patch embedding = the implicit-GEMM conv kernel (14x14 stride 14 on channels-last input), LayerNorms =
``scail_layernorm_affine``, qkv / proj / MLP = ``scail_gemm_bf16`` (GELU-erf and residual epilogues),
attention (16 heads x 80) = ``scail_attn_small``."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from . import lib as L
from . import ops
from .dit import _register


class VisionTransformer(nn.Module):
    def __init__(self, image_size=224, patch_size=14, dim=1280, mlp_ratio=4, out_dim=1024, num_heads=16, num_layers=32,
                 pool_type="token", pre_norm=True, post_norm=False, activation="gelu", norm_eps=1e-5, device=None,
                 init_seed=5, **ignored):
        super().__init__()
        if pool_type != "token" or not pre_norm or post_norm or activation != "gelu":
            raise NotImplementedError("only the clip_xlm_roberta_vit_h_14 visual configuration is implemented (clip.py:460-482)")
        if dim % 64 or (dim // num_heads) % 8:
            raise NotImplementedError("dim must be a multiple of 64 and head_dim a multiple of 8")
        self.image_size, self.patch_size, self.dim = image_size, patch_size, dim
        self.num_heads, self.num_layers, self.norm_eps = num_heads, num_layers, norm_eps
        self.num_patches = (image_size // patch_size) ** 2
        mid = int(dim * mlp_ratio)
        dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        g = torch.Generator(device=dev).manual_seed(init_seed)
        gain = 1.0 / math.sqrt(dim)
        spec = {"patch_embedding.weight": ((dim, 3, patch_size, patch_size), (3 * patch_size ** 2) ** -0.5),
                "cls_embedding": ((1, 1, dim), gain), "pos_embedding": ((1, self.num_patches + 1, dim), gain),
                "pre_norm.weight": ((dim,), "one"), "pre_norm.bias": ((dim,), "zero"),
                "post_norm.weight": ((dim,), "one"), "post_norm.bias": ((dim,), "zero"), "head": ((dim, out_dim), gain)}
        for i in range(num_layers):
            p = f"transformer.{i}."
            for n, shp in (("norm1", dim), ("norm2", dim)):
                spec[p + n + ".weight"] = ((shp,), "one"); spec[p + n + ".bias"] = ((shp,), "zero")
            for n, o, c in (("attn.to_qkv", 3 * dim, dim), ("attn.proj", dim, dim), ("mlp.0", mid, dim), ("mlp.2", dim, mid)):
                spec[p + n + ".weight"] = ((o, c), c ** -0.5); spec[p + n + ".bias"] = ((o,), "zero")
        for n, (shape, std) in spec.items():
            w = torch.ones(shape, device=dev) if std == "one" else (torch.zeros(shape, device=dev) if std == "zero"
                                                                      else torch.randn(shape, device=dev, generator=g) * std)
            _register(self, n, nn.Parameter(w.to(torch.bfloat16), requires_grad=False))
        self._prepared = None

    def load_state_dict(self, *a, **k):
        self._prepared = None
        return super().load_state_dict(*a, **k)

    def prepare(self):
        if self._prepared is None:
            sd = {k: v.detach() for k, v in self.named_parameters()}
            if sd["cls_embedding"].device.type != "cuda":
                raise L.ScailHipError("scail_amd.clip must live on the GPU (no CPU path)")
            L.load()
            f = lambda n: sd[n].float().contiguous()
            m = lambda n: sd[n].to(torch.bfloat16).contiguous()
            W = dict(patch=ops.prep_conv_weight(sd["patch_embedding.weight"], None), cls=m("cls_embedding").view(-1),
                     pos=m("pos_embedding").view(-1, self.dim), pre=(f("pre_norm.weight"), f("pre_norm.bias")), layers=[])
            for i in range(self.num_layers - 1):          # the 32nd block is never evaluated (use_31_block)
                p = f"transformer.{i}."
                W["layers"].append(dict(n1=(f(p + "norm1.weight"), f(p + "norm1.bias")), n2=(f(p + "norm2.weight"), f(p + "norm2.bias")),
                                        qkv=(m(p + "attn.to_qkv.weight"), f(p + "attn.to_qkv.bias")),
                                        proj=(m(p + "attn.proj.weight"), f(p + "attn.proj.bias")),
                                        fc1=(m(p + "mlp.0.weight"), f(p + "mlp.0.bias")), fc2=(m(p + "mlp.2.weight"), f(p + "mlp.2.bias"))))
            self._prepared = W
        return self._prepared

    @torch.no_grad()
    def forward(self, x: torch.Tensor, interpolation=False, use_31_block=True) -> torch.Tensor:
        """x (B, 3, S, S) resized + normalised images -> (B, 1 + (S/patch)^2, dim) bf16 (clip.py:307-326)."""
        if not use_31_block or interpolation:
            raise NotImplementedError("SCAIL only calls visual(..., use_31_block=True) (clip.py:525)")
        W = self.prepare()
        B, _, S, _ = x.shape
        gsz = S // self.patch_size
        if gsz * gsz + 1 != W["pos"].shape[0]:
            raise ValueError("image size does not match the position embedding")
        dev = W["cls"].device
        D, H = self.dim, self.num_heads
        # (B,3,S,S) -> channels-last frames (B, S, S, 8): the B images play the role of the T axis of conv3d_cl
        xcl = ops.to_channels_last(x.to(dev).float().permute(1, 0, 2, 3).contiguous(), 8)
        tok = torch.empty(B, gsz * gsz + 1, D, device=dev, dtype=torch.bfloat16)
        pe = ops.conv3d_cl(xcl, W["patch"], (B, gsz, gsz), stride=(1, self.patch_size, self.patch_size), pad=(0, 0, 0))
        tok[:, 1:].copy_(pe.view(B, gsz * gsz, D))
        tok[:, 0].copy_(W["cls"])
        xs = ops.row_affine(tok, addrow=W["pos"])
        xs = ops.layernorm_affine(xs, *W["pre"], eps=self.norm_eps)
        scale = 1.0 / math.sqrt(D // H)
        for lw in W["layers"]:
            h = ops.layernorm_affine(xs, *lw["n1"], eps=self.norm_eps)
            qkv = ops.gemm(h, *lw["qkv"])
            a = ops.attn_small(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H, scale=scale)
            ops.gemm(a, *lw["proj"], out=xs, epilogue=L.EPI_RESID, resid=xs)
            h = ops.layernorm_affine(xs, *lw["n2"], eps=self.norm_eps, out=h)
            h1 = ops.gemm(h, *lw["fc1"], epilogue=L.EPI_GELU_ERF)
            ops.gemm(h1, *lw["fc2"], out=xs, epilogue=L.EPI_RESID, resid=xs)
        return xs


class _Model(nn.Module):
    def __init__(self, visual):
        super().__init__()
        self.visual = visual


class CLIPModel(nn.Module):
    """clip.py:491-526 (visual tower only: the XLM-R text tower is never used by SCAIL)."""

    MEAN = [0.48145466, 0.4578275, 0.40821073]
    STD = [0.26862954, 0.26130258, 0.27577711]

    def __init__(self, dtype=torch.bfloat16, device="cuda", checkpoint_path=None, **vit_kwargs):
        super().__init__()
        self.model = _Model(VisionTransformer(device=device, **vit_kwargs)).eval()
        if checkpoint_path is not None and __import__("os").path.exists(checkpoint_path):
            sd = torch.load(checkpoint_path, map_location="cpu")
            self.model.load_state_dict({k: v for k, v in sd.items() if k.startswith("visual.")}, strict=False)

    @torch.no_grad()
    def visual(self, videos):
        """videos: iterable of (C, T, H, W) in [-1, 1]; every frame is encoded (clip.py:511-526)."""
        size = (self.model.visual.image_size,) * 2
        x = torch.cat([F.interpolate(u.transpose(0, 1).float(), size=size, mode="bicubic", align_corners=False) for u in videos])
        x = x * 0.5 + 0.5
        mean = torch.tensor(self.MEAN, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(self.STD, device=x.device).view(1, 3, 1, 1)
        return self.model.visual((x - mean) / std, use_31_block=True)

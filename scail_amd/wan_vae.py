"""Wan2.1 causal 3D VAE on MI355X behind the reference's first-stage interface (seam B4).

``WanVAE(z_dim, vae_pth, dtype, device)`` with ``.encode(videos) / .decode(zs)`` and ``.model`` mirrors
``sgm/models/wan_vae.py:619-666``; ``.model`` carries the reference's parameter names so
``Wan2.1_VAE.pth`` loads with ``load_state_dict``.  (The module path contains "wan_vae", which is what
``diffusion_video.py:225-236`` keys the "frozen first stage" branch on.)

Execution differs from the reference by design: the reference streams 1/4/4/... frame chunks through
the network with a 2-frame feature cache per causal conv because a 24-80 GB GPU cannot hold the
activations; an MI355X can (the largest tensor, 96 x 81 x 512 x 896 bf16, is 7 GB of 288 GB), so every
layer runs ONCE over the whole sequence -- no cache clones, no per-chunk launches, no ``torch.cat``
growth -- with the temporal rules of the streamed computation stated explicitly (first-frame bypass of
the temporal down/upsampling convs; see oracle/wan_vae_oracle.py, which is pinned to the chunked
reference).  Activations are channels-last bf16; every conv is the implicit-GEMM kernel of
``csrc/conv.hip``; the residual add is fused into the second conv of each block.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Tuple

import torch
from torch import nn

from . import lib as L
from . import ops
from .dit import _register

LATENT_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
               0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]     # wan_vae.py:630-633
LATENT_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
              3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]           # wan_vae.py:634-637


class WanVAE_(nn.Module):
    """Parameter container + HIP execution of the reference WanVAE_ (wan_vae.py:483-589)."""

    def __init__(self, dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, attn_scales=(),
                 temperal_downsample=(False, True, True), dropout=0.0, device=None, init_seed=4321):
        super().__init__()
        if list(attn_scales):
            raise NotImplementedError("attn_scales is empty in the shipped VAE config (wan_vae.py:602)")
        self.dim, self.z_dim, self.dim_mult = dim, z_dim, tuple(dim_mult)
        self.num_res_blocks = num_res_blocks
        self.temperal_downsample = tuple(temperal_downsample)
        self.temperal_upsample = tuple(temperal_downsample[::-1])
        if dim % 8 or z_dim % 8:
            raise NotImplementedError("dim and z_dim must be multiples of 8")
        dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        g = torch.Generator(device=dev).manual_seed(init_seed)
        for n, shape in self.param_spec().items():
            if n.endswith("gamma"):
                w = torch.ones(shape, device=dev)
            elif n.endswith("bias"):
                w = torch.zeros(shape, device=dev)
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                w = torch.randn(shape, device=dev, generator=g) / math.sqrt(fan_in)
            _register(self, n, nn.Parameter(w.to(torch.bfloat16), requires_grad=False))
        self._prepared = None
        # one C call per encode / decode (include/scail_vae.h); the layer-by-layer path below is the cross-check
        self.use_c_exec = os.environ.get("SCAIL_C_VAE", "1") != "0"
        self._cvae = None

    # ---- architecture tables (Encoder3d :283-306, Decoder3d :387-416) ----
    def encoder_plan(self):
        dims = [self.dim * u for u in (1,) + self.dim_mult]
        plan, idx = [], 0
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(self.num_res_blocks):
                plan.append(("res", f"encoder.downsamples.{idx}", cin, cout)); idx += 1
                cin = cout
            if i != len(self.dim_mult) - 1:
                plan.append(("down", f"encoder.downsamples.{idx}", cout, self.temperal_downsample[i])); idx += 1
        return plan

    def decoder_plan(self):
        dims = [self.dim * u for u in (self.dim_mult[-1],) + self.dim_mult[::-1]]
        plan, idx = [], 0
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            if i in (1, 2, 3):
                cin = cin // 2
            for _ in range(self.num_res_blocks + 1):
                plan.append(("res", f"decoder.upsamples.{idx}", cin, cout)); idx += 1
                cin = cout
            if i != len(self.dim_mult) - 1:
                plan.append(("up", f"decoder.upsamples.{idx}", cout, self.temperal_upsample[i])); idx += 1
        return plan

    def param_spec(self) -> Dict[str, Tuple[int, ...]]:
        s: Dict[str, Tuple[int, ...]] = {}

        def conv3(n, co, ci, k):
            s[n + ".weight"] = (co, ci, *k); s[n + ".bias"] = (co,)

        def res(n, ci, co):
            s[n + ".residual.0.gamma"] = (ci, 1, 1, 1)
            conv3(n + ".residual.2", co, ci, (3, 3, 3))
            s[n + ".residual.3.gamma"] = (co, 1, 1, 1)
            conv3(n + ".residual.6", co, co, (3, 3, 3))
            if ci != co:
                conv3(n + ".shortcut", co, ci, (1, 1, 1))

        def attn(n, c):
            s[n + ".norm.gamma"] = (c, 1, 1)
            s[n + ".to_qkv.weight"] = (3 * c, c, 1, 1); s[n + ".to_qkv.bias"] = (3 * c,)
            s[n + ".proj.weight"] = (c, c, 1, 1); s[n + ".proj.bias"] = (c,)

        d, z = self.dim, self.z_dim
        top = d * self.dim_mult[-1]
        conv3("encoder.conv1", d, 3, (3, 3, 3))
        for kind, n, a, b in self.encoder_plan():
            if kind == "res":
                res(n, a, b)
            else:
                s[n + ".resample.1.weight"] = (a, a, 3, 3); s[n + ".resample.1.bias"] = (a,)
                if b:
                    conv3(n + ".time_conv", a, a, (3, 1, 1))
        res("encoder.middle.0", top, top); attn("encoder.middle.1", top); res("encoder.middle.2", top, top)
        s["encoder.head.0.gamma"] = (top, 1, 1, 1)
        conv3("encoder.head.2", 2 * z, top, (3, 3, 3))
        conv3("conv1", 2 * z, 2 * z, (1, 1, 1))
        conv3("conv2", z, z, (1, 1, 1))
        conv3("decoder.conv1", top, z, (3, 3, 3))
        res("decoder.middle.0", top, top); attn("decoder.middle.1", top); res("decoder.middle.2", top, top)
        for kind, n, a, b in self.decoder_plan():
            if kind == "res":
                res(n, a, b)
            else:
                s[n + ".resample.1.weight"] = (a // 2, a, 3, 3); s[n + ".resample.1.bias"] = (a // 2,)
                if b:
                    conv3(n + ".time_conv", 2 * a, a, (3, 1, 1))
        s["decoder.head.0.gamma"] = (d, 1, 1, 1)
        conv3("decoder.head.2", 3, d, (3, 3, 3))
        return s

    def load_state_dict(self, *a, **k):
        self._prepared = None
        return super().load_state_dict(*a, **k)

    # ---- kernel-side weights ----
    def prepare(self):
        if self._prepared is not None:
            return self._prepared
        sd = {k: v.detach() for k, v in self.named_parameters()}
        if next(iter(sd.values())).device.type != "cuda":
            raise L.ScailHipError("scail_amd.wan_vae must live on the GPU (no CPU path)")
        L.load()
        W = {}
        for n in sd:
            if n.endswith(".weight") and not n.endswith("to_qkv.weight") and not n.endswith("proj.weight"):
                base = n[:-7]
                if base.endswith("time_conv") and sd[n].shape[0] == 2 * sd[n].shape[1]:      # upsample3d: two output halves
                    c = sd[n].shape[1]
                    W[base + "#0"] = ops.prep_conv_weight(sd[n][:c], sd[base + ".bias"][:c])
                    W[base + "#1"] = ops.prep_conv_weight(sd[n][c:], sd[base + ".bias"][c:])
                else:
                    W[base] = ops.prep_conv_weight(sd[n], sd[base + ".bias"])
            elif n.endswith("gamma"):
                W[n] = sd[n].float().reshape(-1).contiguous()
            elif n.endswith("to_qkv.weight"):
                base = n[:-14]
                c = sd[n].shape[1]
                w = sd[n].reshape(3 * c, c).to(torch.bfloat16)
                b = sd[base + ".to_qkv.bias"].float()
                for j, nm in enumerate("qkv"):
                    W[base + "." + nm] = (w[j * c:(j + 1) * c].contiguous(), b[j * c:(j + 1) * c].contiguous())
                W[base + ".proj"] = (sd[base + ".proj.weight"].reshape(c, c).to(torch.bfloat16).contiguous(),
                                     sd[base + ".proj.bias"].float().contiguous())
        dev = next(iter(sd.values())).device
        W["mean"] = torch.tensor(LATENT_MEAN[:self.z_dim], device=dev)
        W["std"] = torch.tensor(LATENT_STD[:self.z_dim], device=dev)
        self._prepared = W
        self._cvae = None
        return W

    def _c(self):
        if self._cvae is None:
            from .cvae import CVae
            self._cvae = CVae(self, self.prepare())
        return self._cvae

    # ---- blocks (channels-last (T,H,W,C) bf16) ----
    def _res(self, W, n, x, xn=None, next_gamma=None, need_raw=True):
        """ResidualBlock (wan_vae.py:180-218).  xn: SiLU(RMS_norm(x) * residual.0.gamma) when x's producer already wrote it.  next_gamma: the norm
        weights of whatever reads this block's output next; where ONE generated kernel covers "last convolution + shortcut sum + that norm"
        (ops.conv_resid_norm_generated, the rule of csrc/vae_exec.hip res_block) the block returns (raw | None, normalised), else (raw, None)."""
        T, H, Wd, _ = x.shape
        h = ops.conv3d_cl(x, W[n + ".shortcut"], (T, H, Wd)) if (n + ".shortcut") in W else x
        y = xn if xn is not None else ops.rms_silu(x, W[n + ".residual.0.gamma"])
        if ops.conv_norm_fusable(W[n + ".residual.2"], y.shape[3]) and (ops.conv_norm_generated(W[n + ".residual.2"], y.shape) or
                                                                         not ops.conv_generated(W[n + ".residual.2"], y.shape)):
            # conv -> RMS_norm -> SiLU in one kernel: the generated kernel's norm epilogue (96 channels), or the hipcc halo kernel's for the
            # shapes the generated kernels do not cover (same rule as csrc/vae_exec.hip res_block)
            y = ops.conv3d_cl_norm(y, W[n + ".residual.2"], W[n + ".residual.3.gamma"])
        else:
            y = ops.conv3d_cl(y, W[n + ".residual.2"], (T, H, Wd))
            ops.rms_silu(y, W[n + ".residual.3.gamma"], out=y)
        w6 = W[n + ".residual.6"]
        if next_gamma is not None and tuple(w6["k"]) == (3, 3, 3) and ops.conv_resid_norm_generated(w6, y.shape) and h.shape[3] == w6["N"]:
            return ops.conv3d_cl_resid_norm(y, w6, h, next_gamma, want_raw=need_raw)
        return ops.conv3d_cl(y, w6, (T, H, Wd), resid=h), None

    def _stages(self, W, plan, x, down, head_gamma=None, xn=None):
        """the stage table of the encoder / decoder; a ResidualBlock followed by another one (or, last in the decoder, by the head's norm) hands
        its consumer's normalised input along.  Returns (raw | None, normalised | None)."""
        for i, (kind, n, a, b) in enumerate(plan):
            if kind == "res":
                last = i + 1 == len(plan)
                nxt = head_gamma if (last and head_gamma is not None) else (W[plan[i + 1][1] + ".residual.0.gamma"] if (not last and plan[i + 1][0] == "res") else None)
                x, xn = self._res(W, n, x, xn, nxt, need_raw=not (last and head_gamma is not None))
            elif down:
                x = self._down(W, n, x, b)
            else:
                nxt = W[plan[i + 1][1] + ".residual.0.gamma"] if (i + 1 < len(plan) and plan[i + 1][0] == "res") else None
                x, xn = self._up(W, n, x, b, nxt) if nxt is not None else (self._up(W, n, x, b), None)
        return x, xn

    def _attn(self, W, n, x):
        T, H, Wd, C = x.shape
        nt = H * Wd
        if nt > 32768:
            raise NotImplementedError(f"mid-block attention handles up to 32768 tokens per frame, got (H/8)*(W/8) = {nt}")
        y = ops.rms_silu(x, W[n + ".norm.gamma"], silu=False).view(T * nt, C)
        kbuf = torch.zeros(T * nt + 8, C, device=x.device, dtype=torch.bfloat16)    # + 8 rows: score GEMM reads up to ceil8(nt) keys
        q, v = (ops.gemm(y, *W[n + "." + nm]) for nm in "qv")
        ops.gemm(y, *W[n + ".k"], out=kbuf[:T * nt])
        npad, nt8 = (nt + 63) // 64 * 64, (nt + 7) // 8 * 8
        S = torch.zeros(nt, npad, device=x.device, dtype=torch.bfloat16)
        vt = torch.zeros(1, C, npad, device=x.device, dtype=torch.bfloat16)
        o = torch.empty(T * nt, C, device=x.device, dtype=torch.bfloat16)
        for f in range(T):
            sl = slice(f * nt, (f + 1) * nt)
            ops.gemm(q[sl], kbuf[f * nt:f * nt + nt8], out=S[:, :nt8])      # the <= 7 extra columns are zeroed by the softmax
            ops.softmax_rows_(S, nt, 1.0 / math.sqrt(C))
            ops.transpose2d(v[sl].unsqueeze(0), vt)
            ops.gemm(S, vt[0], out=o[sl])
        wp, bp = W[n + ".proj"]
        out = ops.gemm(o, wp, bp, epilogue=L.EPI_RESID, resid=x.view(T * nt, C))
        return out.view(T, H, Wd, C)

    def _down(self, W, n, x, temporal):
        T, H, Wd, C = x.shape
        y = ops.conv3d_cl(x, W[n + ".resample.1"], (T, H // 2, Wd // 2), stride=(1, 2, 2), pad=(0, 0, 0))
        if temporal and T > 1:
            To = (T - 1) // 2
            out = torch.empty(1 + To, H // 2, Wd // 2, C, device=x.device, dtype=torch.bfloat16)
            out[0].copy_(y[0])                                     # first frame bypasses the temporal conv (:146-148)
            ops.conv3d_cl(y, W[n + ".time_conv"], (To, H // 2, Wd // 2), stride=(2, 1, 1), pad=(0, 0, 0), out=out, ot_off=1)
            y = out
        return y

    def _up(self, W, n, x, temporal, next_gamma=None):
        """Resample upsample2d / upsample3d; with next_gamma (the next ResidualBlock's residual.0 norm) returns (raw, normalised | None)."""
        T, H, Wd, C = x.shape
        if temporal and T > 1:
            t2 = torch.empty(1 + 2 * (T - 1), H, Wd, C, device=x.device, dtype=torch.bfloat16)
            t2[0].copy_(x[0])                                      # 'Rep': first latent frame is not doubled (:106-108)
            tail = x[1:]                                           # frames >= 1 never see frame 0 (:120-130)
            for p in (0, 1):
                ops.conv3d_cl(tail, W[n + f".time_conv#{p}"], (T - 1, H, Wd), out=t2, ot_mul=2, ot_off=1 + p)
            x, T = t2, t2.shape[0]
        wr = W[n + ".resample.1"]
        if next_gamma is not None and ops.conv_resid_norm_generated(wr, x.shape, (T, 2 * H, 2 * Wd), pad=(0, 1, 1), ups=True, resid=False):
            return ops.conv3d_cl_resid_norm(x, wr, None, next_gamma, out_shape=(T, 2 * H, 2 * Wd), pad=(0, 1, 1), ups=True)
        y = ops.conv3d_cl(x, wr, (T, 2 * H, 2 * Wd), pad=(0, 1, 1), ups=True)
        return (y, None) if next_gamma is not None else y

    # ---- public: one video / one latent, planar fp32 in and out (reference layout) ----
    @torch.no_grad()
    def encode(self, video: torch.Tensor, scale=None) -> torch.Tensor:
        """video (1|-, 3, T, H, W) -> normalised mean (1, z, 1+(T-1)/4, H/8, W/8) fp32 (wan_vae.py:516-542)."""
        if video.dim() == 5:
            assert video.shape[0] == 1
            video = video[0]
        C3, T, H, Wd = video.shape
        if (T - 1) % 4 or H % 8 or Wd % 8:
            raise ValueError("video needs T = 1 + 4n frames and H, W multiples of 8")
        W = self.prepare()
        if self.use_c_exec:
            return self._c().encode(video.float().to(next(self.parameters()).device).contiguous()).unsqueeze(0)
        x = ops.to_channels_last(video.float().to(next(self.parameters()).device), 8)
        plan = self.encoder_plan()
        g0 = W[plan[0][1] + ".residual.0.gamma"] if plan and plan[0][0] == "res" else None
        xn = None
        if g0 is not None and ops.conv_resid_norm_generated(W["encoder.conv1"], x.shape, resid=False):
            x, xn = ops.conv3d_cl_resid_norm(x, W["encoder.conv1"], None, g0)          # the stem with the first block's input norm (csrc/vae_exec.hip)
        else:
            x = ops.conv3d_cl(x, W["encoder.conv1"], (T, H, Wd))
        x, _ = self._stages(W, plan, x, down=True, xn=xn)
        x, _ = self._res(W, "encoder.middle.0", x)
        x = self._attn(W, "encoder.middle.1", x)
        x, _ = self._res(W, "encoder.middle.2", x)
        ops.rms_silu(x, W["encoder.head.0.gamma"], out=x)
        x = ops.conv3d_cl(x, W["encoder.head.2"], x.shape[:3])
        x = ops.conv3d_cl(x, W["conv1"], x.shape[:3])
        mu = ops.from_channels_last(x, self.z_dim, a=(1.0 / W["std"]).contiguous(), b=(-W["mean"]).contiguous())
        return mu.unsqueeze(0)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, scale=None) -> torch.Tensor:
        """z (1|-, zc, T, h, w) -> video (1, 3, 1+4(T-1), 8h, 8w) fp32, NOT clamped (wan_vae.py:544-568)."""
        if z.dim() == 5:
            assert z.shape[0] == 1
            z = z[0]
        W = self.prepare()
        if self.use_c_exec:
            return self._c().decode(z.float().to(next(self.parameters()).device).contiguous()).unsqueeze(0)
        x = ops.to_channels_last(z.float().to(next(self.parameters()).device), self.z_dim, a=W["std"], b=W["mean"])
        x = ops.conv3d_cl(x, W["conv2"], x.shape[:3])
        x = ops.conv3d_cl(x, W["decoder.conv1"], x.shape[:3])
        x, _ = self._res(W, "decoder.middle.0", x)
        x = self._attn(W, "decoder.middle.1", x)
        x, _ = self._res(W, "decoder.middle.2", x)
        x, xn = self._stages(W, self.decoder_plan(), x, down=False, head_gamma=W["decoder.head.0.gamma"])
        if xn is not None:          # the head's RMS_norm + SiLU came out of the last block's epilogue
            x = xn
        else:
            ops.rms_silu(x, W["decoder.head.0.gamma"], out=x)
        x = ops.conv3d_cl(x, W["decoder.head.2"], x.shape[:3])
        return ops.from_channels_last(x, 3).unsqueeze(0)


class WanVAE:
    """Reference-compatible wrapper (wan_vae.py:619-666)."""

    def __init__(self, z_dim=16, vae_pth=None, dtype=torch.bfloat16, device="cuda", dim=96, **kwargs):
        dtype = eval(dtype) if isinstance(dtype, str) else dtype           # the reference yaml passes "torch.bfloat16"
        self.dtype, self.device = dtype, device
        self.mean = torch.tensor(LATENT_MEAN[:z_dim], dtype=torch.float32, device=device)
        self.std = torch.tensor(LATENT_STD[:z_dim], dtype=torch.float32, device=device)
        self.scale = [self.mean, 1.0 / self.std]
        self.model = WanVAE_(dim=dim, z_dim=z_dim, device=device).eval().requires_grad_(False)
        if vae_pth is not None:
            if not os.path.exists(vae_pth):
                raise FileNotFoundError(f"vae_pth {vae_pth} does not exist (pass vae_pth=None for random-init weights)")
            from .checkpoint import load_vae_pth
            load_vae_pth(self.model, vae_pth, device=device)        # strict, like wan_vae.py:607-616

    def encode(self, videos):
        """videos: iterable of [C, T, H, W] -> latent float (wan_vae.py:648-657)."""
        return torch.cat([self.model.encode(u.unsqueeze(0)).float() for u in videos], dim=0)

    def decode(self, zs):
        """zs: iterable of [z, T, h, w] -> float video clamped to [-1, 1] (wan_vae.py:659-666)."""
        return torch.cat([self.model.decode(u.unsqueeze(0)).float().clamp_(-1, 1) for u in zs], dim=0)

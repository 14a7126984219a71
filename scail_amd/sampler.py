"""sgm sampler stack for the SCAIL hot path, mirrored class-for-class so the reference's yaml
``target:`` strings can be pointed here (seam between L5 engine and L3 DiT).

  RFSampler          sgm/modules/diffusionmodules/sampling.py:920-982 (+ make_flow_timesteps :888-903)
  RFSamplerLong      sampling.py:986-1085 (temporal tiling for videos longer than the trained window)
  Denoiser/RFScaling sgm/modules/diffusionmodules/denoiser.py:9-43, denoiser_scaling.py:71-78
  VanillaCFG         sgm/modules/diffusionmodules/guiders.py:23-57 (+ sampling_utils.py:7-10)
  OpenAIWrapper      sgm/modules/diffusionmodules/wrappers.py:24-45

The generic path keeps the reference call protocol (``sampler(denoiser, x, cond, uc)``,
``denoiser(network, input, sigma, cond, **kw)``, ``network(x, t, c, **kw)``) and works with any
network.  When the network is a ``scail_amd.dit.DiffusionTransformer`` the sampler step uses the
fused HIP kernel for CFG-combine + Euler update (``scail_cfg_euler``) on the fp32 state instead of
four elementwise torch ops; the arithmetic is the same fp32 expression.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
from torch import nn

from . import ops


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    """sgm/util.py:292 -- append trailing singleton dims."""
    return x[(...,) + (None,) * (target_dims - x.ndim)]


def make_flow_timesteps(t_start, num_flow_steps, verbose=False, shift_scale=7, mode="normal"):
    """sampling.py:888-903: linspace in float64, hunyuan shift, float32, 1 - s for mode 'normal'."""
    s = np.linspace(t_start, 1.0, num_flow_steps + 1, endpoint=True)
    s = s / (shift_scale + s - shift_scale * s)
    s = torch.tensor(s, dtype=torch.float32)
    if mode == "normal":
        s = 1 - s
    elif mode != "meta":
        raise ValueError(f"Unknown mode {mode}.")
    if verbose:
        print(f"Selected timesteps for flow sampler: {s}")
    return s


class RFScaling:
    """denoiser_scaling.py:71-78."""

    def __call__(self, sigma, **additional_model_inputs):
        return torch.zeros_like(sigma), torch.ones_like(sigma), torch.ones_like(sigma), sigma.clone() * 1000


class EpsWeighting:
    """denoiser_weighting.py (training only; kept so denoiser_config instantiates)."""

    def __call__(self, sigma):
        return sigma ** -2.0


class Denoiser(nn.Module):
    """denoiser.py:9-43 with scaling_config resolved to RFScaling by default."""

    def __init__(self, weighting_config=None, scaling_config=None):
        super().__init__()
        from .config import instantiate_from_config
        self.weighting = instantiate_from_config(weighting_config) if weighting_config else EpsWeighting()
        self.scaling = instantiate_from_config(scaling_config) if scaling_config else RFScaling()

    def forward(self, network, input, sigma, cond: Dict, **additional_model_inputs):
        sigma_shape = sigma.shape
        sigma = append_dims(sigma, input.ndim)
        c_skip, c_out, c_in, c_noise = self.scaling(sigma, **additional_model_inputs)
        c_noise = c_noise.reshape(sigma_shape)
        model_output = network(input * c_in, c_noise, cond, **additional_model_inputs)
        return model_output * c_out + input * c_skip


class NoDynamicThresholding:
    """sampling_utils.py:7-10."""

    def __call__(self, uncond, cond, scale):
        scale = append_dims(scale, cond.ndim) if isinstance(scale, torch.Tensor) else scale
        return uncond + scale * (cond - uncond)


class VanillaCFG:
    """guiders.py:23-57: batch-2 (uncond, cond) assembly and u + s (c - u)."""

    def __init__(self, scale, dyn_thresh_config=None):
        self.scale = scale
        self.dyn_thresh = NoDynamicThresholding()

    def __call__(self, x, sigma, scale=None):
        x_u, x_c = x.chunk(2)
        return self.dyn_thresh(x_u, x_c, self.scale if scale is None else scale)

    def prepare_inputs(self, x, s, c, uc):
        c_out = dict()
        for k in c:
            if k in ["vector", "crossattn", "concat"]:
                if uc[k].shape[1] != c[k].shape[1]:
                    uc[k] = torch.cat([uc[k], uc[k][:, -1:].repeat(1, abs(c[k].shape[1] - uc[k].shape[1]), 1)], dim=1)
                c_out[k] = torch.cat((uc[k], c[k]), 0)
            else:
                c_out[k] = c[k]
        # the batch below is ONE latent and ONE sigma twice: scail_amd's network evaluates layer 0 up to its first cross attention once
        # (include/scail_dit.h SCAIL_DIT_CFG_PAIR; result-preserving).  A plain bool: other networks ignore the key.
        c_out["cfg_pair"] = True
        return torch.cat([x] * 2), torch.cat([s] * 2), c_out


class OpenAIWrapper(nn.Module):
    """wrappers.py:24-45 (IdentityWrapper + OpenAIWrapper)."""

    def __init__(self, diffusion_model, compile_model: bool = False, dtype=torch.float32, **kwargs):
        super().__init__()
        self.diffusion_model = diffusion_model
        self.dtype = dtype

    def forward(self, x, t, c: dict, **kwargs):
        for key in c:
            if torch.is_tensor(c[key]):
                c[key] = c[key].to(self.dtype)
        kwargs.update(c)
        if "concat" in c:
            x = torch.cat((x, c["concat"]), dim=2 if x.dim() == 5 else 1)
        return self.diffusion_model(x, timesteps=t, context=c.get("crossattn", None), y=c.get("vector", None), **kwargs)


class RFDiscretization:
    """discretizer.py:131-180 is bypassed by hunyuan_schedule=True in the shipped config
    (sampling.py:941-942); kept as a named target so sampler_config instantiates."""

    def __init__(self, reverse=False, **kw):
        self.reverse = reverse


class RFSampler:
    """sampling.py:920-982.  Only the shipped branch (hunyuan_schedule) is implemented.

    One deliberate difference: a ``num_steps`` ARGUMENT (``__call__`` / ``sample_hip`` / ``engine.sample``) overrides the configured
    step count here.  In the reference the argument only reaches the discretization, whose result the hunyuan schedule then
    replaces with ``make_flow_timesteps(0, self.num_steps, ...)`` (:936-942) -- i.e. it is silently ignored; the reference's own
    engine never passes it (diffusion_video.py:565-569).  Leave it ``None`` for the reference's behaviour."""

    def __init__(self, schedule_shift=False, hunyuan_schedule=False, shift_scale=7, mode="normal", distill=False,
                 discretization_config=None, num_steps=None, guider_config=None, verbose=False, device="cuda"):
        from .config import instantiate_from_config
        if schedule_shift or not hunyuan_schedule or distill:
            raise NotImplementedError("only schedule_shift=False, hunyuan_schedule=True, distill=False (shipped config)")
        self.num_steps, self.shift_scale, self.mode, self.verbose, self.device = num_steps, shift_scale, mode, verbose, device
        self.guider = instantiate_from_config(guider_config) if guider_config else VanillaCFG(1.0)

    def sigmas(self, num_steps=None) -> torch.Tensor:
        return make_flow_timesteps(0, self.num_steps if num_steps is None else num_steps, verbose=False,
                                   shift_scale=self.shift_scale, mode=self.mode)

    def denoise(self, x, denoiser, sigma, cond, uc, scale=None, fps=None):
        """sampling.py:950-958."""
        extra = {"cfg_scale": scale if scale is not None else self.guider.scale}
        denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc), **extra).to(torch.float32)
        return self.guider(denoised, sigma)

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, scale=None, fps=None):
        """sampling.py:960-963."""
        output = self.denoise(x, denoiser, sigma, cond, uc, scale=scale, fps=fps).to(torch.float32)
        return x + append_dims(next_sigma - sigma, x.ndim) * output

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, scale=None, ofs=None, fps=None):
        sigmas = self.sigmas(num_steps).to(x.device)
        uc = cond if uc is None else uc
        s_in = x.new_ones([x.shape[0]])
        for i in range(len(sigmas) - 1):
            x = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc, scale=scale, fps=fps)
        return x

    # ---- fused path for the HIP network (same arithmetic, fewer passes / no host syncs) ----
    def sample_hip(self, network, x, cond: Dict, uc: Dict, num_steps=None, scale=None, chunk_dim=None,
                   step_callback=None):
        """x (1,T,16,H,W) fp32 on the GPU; cond/uc as the CLI builds them (sample_video.py:455-470).
        One step = one batch-2 DiT forward (uncond, cond) + scail_cfg_euler."""
        sig = self.sigmas(num_steps)                     # host fp32, like the reference
        cfg = float(self.guider.scale if scale is None else scale)
        ctx = torch.cat((uc["crossattn"], cond["crossattn"]), 0)
        shared = {k: v for k, v in cond.items() if k != "crossattn"}
        if (getattr(network, "use_c_step", False) and step_callback is None and chunk_dim is None
                and getattr(network, "kernel_timer", None) is None and getattr(network, "_tap", None) is None
                and (network.sp is None or network.sp.size == 1)
                and shared["ref_concat"].shape[0] == 1 and shared["concat_smpl_render"].shape[0] == 1
                and shared["ref_concat"].shape[1] == 1):
            # the whole loop enqueued by ONE call into the library (scail_dit_sample); same kernels, same order
            return network.sample_c(x, sig, cfg, ctx, shared["ref_concat"], shared["concat_smpl_render"],
                                    shared["image_clip_features"], cond_key=("sample_hip", id(cond)))
        x = x.float().contiguous().clone()
        for i in range(len(sig) - 1):
            xin = torch.cat([x, x], 0)
            t = (sig[i] * 1000.0).repeat(2).to(x.device)                 # c_noise = 1000 sigma (RFScaling)
            v = network.forward_f32(xin, t, ctx, None, cond_key=("sample_hip", id(cond)), chunk_dim=chunk_dim, **shared)
            ops.cfg_euler_(x, v, cfg, float(sig[i + 1] - sig[i]))
            if step_callback is not None:
                step_callback(i, x)
        return x


class RFSamplerLong(RFSampler):
    """sampling.py:986-1085: every step denoises overlapping temporal tiles of the latent (``tile_indices``: lists of
    frame indices, equal length) against the matching pose tile ``cond['smpl_tiled'][:, k]`` and blends the
    CFG-combined predictions with triangular weights before one Euler update of the whole latent.

    The reference walks the pairs (k, k+1) and therefore denoises every interior tile twice with identical inputs;
    here each tile is evaluated once and enters the weighted sums with the same multiplicity (x2 is exact in
    floating point), which halves the network evaluations for long videos.  Needs >= 2 tiles like the reference
    (a single tile leaves its weight_sum at zero)."""

    @staticmethod
    def tile_weight(n: int, device=None) -> torch.Tensor:
        w = (torch.arange(n, device=device, dtype=torch.float32) + 0.5) * 2.0 / n
        return torch.minimum(w, 2.0 - w)

    @staticmethod
    def _mult(k: int, n: int) -> int:
        return 1 if (k == 0 or k == n - 1) else 2

    @staticmethod
    def _check(tile_indices):
        if tile_indices is None or len(tile_indices) < 2:
            raise ValueError("RFSamplerLong needs tile_indices with at least two temporal tiles")
        n0 = len(tile_indices[0])
        if any(len(t) != n0 for t in tile_indices):
            raise ValueError("all temporal tiles must have the same length")

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, scale=None, fps=None, tile_indices=None,
                     smpl_tiled=None):
        """sampling.py:1036-1068 (generic protocol: any denoiser / network)."""
        self._check(tile_indices)
        n = len(tile_indices)
        denoised = torch.zeros_like(x)
        weight_sum = torch.zeros((x.shape[1],), device=x.device)
        weight = self.tile_weight(len(tile_indices[0]), x.device)
        for k in range(n):
            idx = list(tile_indices[k])
            c_k, uc_k = dict(cond), dict(uc)
            c_k["concat_smpl_render"] = smpl_tiled[:, k]
            uc_k["concat_smpl_render"] = smpl_tiled[:, k]
            d = self.denoise(x[:, idx], denoiser, sigma, c_k, uc_k, scale=scale, fps=fps).to(torch.float32)
            m = self._mult(k, n)
            denoised[:, idx] += m * d * weight[:, None, None, None]
            weight_sum[idx] += m * weight
        denoised.div_(weight_sum[:, None, None, None])
        return x + append_dims(next_sigma - sigma, x.ndim) * denoised

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, scale=None, ofs=None, fps=None, tile_indices=None):
        sigmas = self.sigmas(num_steps).to(x.device)
        uc = cond if uc is None else uc
        s_in = x.new_ones([x.shape[0]])
        smpl_tiled = cond["smpl_tiled"]
        for i in range(len(sigmas) - 1):
            x = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc, scale=scale, fps=fps,
                                  tile_indices=tile_indices, smpl_tiled=smpl_tiled)
        return x

    def sample_hip(self, network, x, cond: Dict, uc: Dict, num_steps=None, scale=None, chunk_dim=None,
                   step_callback=None, tile_indices=None):
        """Fused path for the HIP network: per step and tile one batch-2 DiT forward (fp32 out); the text / CLIP
        conditioning (step- AND tile-invariant) is computed once per request."""
        self._check(tile_indices)
        n = len(tile_indices)
        sig = self.sigmas(num_steps)
        cfg = float(self.guider.scale if scale is None else scale)
        ctx = torch.cat((uc["crossattn"], cond["crossattn"]), 0)
        shared = {k: v for k, v in cond.items() if k not in ("crossattn", "smpl_tiled", "concat_smpl_render")}
        smpl_tiled = cond["smpl_tiled"]
        x = x.float().contiguous().clone()
        dev = x.device
        weight = self.tile_weight(len(tile_indices[0]), dev)[:, None, None, None]
        idxs = [torch.as_tensor(list(t), device=dev, dtype=torch.long) for t in tile_indices]
        wsum = torch.zeros(x.shape[1], device=dev)
        for k in range(n):
            wsum[idxs[k]] += self._mult(k, n) * weight[:, 0, 0, 0]
        inv = (1.0 / wsum)[:, None, None, None]
        for i in range(len(sig) - 1):
            t = (sig[i] * 1000.0).repeat(2).to(dev)
            den = torch.zeros_like(x)
            for k in range(n):
                xt = x[:, idxs[k]]
                v = network.forward_f32(torch.cat([xt, xt], 0), t, ctx, None, cond_key=("sample_hip", id(cond)),
                                        chunk_dim=chunk_dim, concat_smpl_render=smpl_tiled[:, k], **shared)
                d = v[0:1] + cfg * (v[1:2] - v[0:1])                     # VanillaCFG, guiders.py:41-45
                den[:, idxs[k]] += (self._mult(k, n) * weight) * d
            x = x + float(sig[i + 1] - sig[i]) * (den * inv)
            if step_callback is not None:
                step_callback(i, x)
        return x

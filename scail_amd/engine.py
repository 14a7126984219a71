"""Sampling engine: the ``sample`` / first-stage surface of SATVideoDiffusionEngine
(reference diffusion_video.py:41-174, :298-331, :456-587) on top of the HIP network.

Only the inference members the CLI uses are reproduced: ``sample(cond, uc, batch_size, shape, ...)``,
``encode_first_stage`` / ``decode_first_stage``, ``.model`` (OpenAIWrapper), ``.denoiser``, ``.sampler``, and -- opt-in --
``.conditioner`` (scail_amd/conditioner.py) and ``.i2v_clip``.
Training (``shared_step``, loss, EMA) is out of scope."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import nn

from . import sampler as S
from .config import instantiate_from_config
from .dit import DiffusionTransformer


class SATVideoDiffusionEngine(nn.Module):
    def __init__(self, model_config: Dict, device="cuda", seed_offset: int = 0, sp=None):
        super().__init__()
        net_cfg = dict(model_config["network_config"])
        params = dict(net_cfg.get("params", {}))
        params["dtype"] = "bf16"                                           # diffusion_video.py:92-105
        params["use_i2v_clip"] = model_config.get("use_i2v_clip", False)
        params["device"] = device
        net = instantiate_from_config({"target": net_cfg["target"], "params": params})
        self.model = S.OpenAIWrapper(net, dtype=torch.bfloat16)
        self.denoiser = instantiate_from_config(model_config["denoiser_config"]) if "denoiser_config" in model_config else S.Denoiser()
        scfg = dict(model_config["sampler_config"])
        sp_params = dict(scfg.get("params", {}))
        sp_params["device"] = device
        self.sampler = instantiate_from_config({"target": scfg["target"], "params": sp_params})
        self.scale_factor = model_config.get("scale_factor", 1.0)
        self.latent_input = model_config.get("latent_input", False)
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self.first_stage_model = None
        self.sp = sp
        net.sp = sp
        fs = model_config.get("first_stage_config")
        if fs is not None and model_config.get("build_first_stage", False):
            self.first_stage_model = instantiate_from_config(fs)
        # request-time encoders (diffusion_video.py:113-127: conditioner + i2v_clip).  The reference always builds them; here
        # they are opt-in because their checkpoints / tokenizer files do not exist offline and they cost 13 GB of HBM.
        self.use_i2v_clip = model_config.get("use_i2v_clip", False)
        self.conditioner = self.i2v_clip = None
        if model_config.get("build_conditioner", False) and "conditioner_config" in model_config:
            self.conditioner = instantiate_from_config(model_config["conditioner_config"])
        if model_config.get("build_i2v_clip", False) and "i2v_clip_config" in model_config:
            self.i2v_clip = instantiate_from_config(model_config["i2v_clip_config"])

    @property
    def network(self) -> DiffusionTransformer:
        return self.model.diffusion_model

    @torch.no_grad()
    def sample(self, cond: Dict, uc: Optional[Dict] = None, batch_size: int = 1, shape: Union[None, Tuple, List] = None,
               prefix=None, concat_images=None, ofs=None, fps=None, tile_indices=None,
               generator: Optional[torch.Generator] = None, num_steps: Optional[int] = None, fused: bool = True, **kwargs):
        """diffusion_video.py:456-587.  shape = (T, C, H, W).  Noise is drawn in fp32 on the host RNG
        stream of ``generator`` like ``torch.randn(batch, *shape)`` (:470), broadcast inside the
        sequence-parallel group (:486-493) and H/W-chunked (:495-552); result gathered to SP rank 0 (:571-585)."""
        randn = torch.randn(batch_size, *shape, generator=generator).to(torch.float32).to(self.device)
        if prefix is not None:
            randn = torch.cat([prefix, randn[:, prefix.shape[1]:]], dim=1)
        chunk_dim = None
        sp = self.sp if (self.sp is not None and self.sp.size > 1) else None
        uc = cond if uc is None else uc
        if sp is not None:
            sp.broadcast(randn)
            h, w = shape[-2:]
            chunk_dim = 3 if h < w else 4
            randn = sp.chunk(randn, chunk_dim)
            cond, uc = dict(cond), dict(uc)
            for k in ("concat_images", "ref_concat", "concat_pose", "concat_smpl_render"):
                if k in cond and cond[k].dim() > chunk_dim:                # a placeholder concat_images (never read) has no such axis
                    cond[k] = sp.chunk(cond[k], chunk_dim)
                    uc[k] = sp.chunk(uc[k], chunk_dim)
            if "smpl_tiled" in cond:                                       # one more leading (tile) axis, :518-524
                cond["smpl_tiled"] = sp.chunk(cond["smpl_tiled"], chunk_dim + 1)
                uc["smpl_tiled"] = sp.chunk(uc["smpl_tiled"], chunk_dim + 1)
        extra = {} if tile_indices is None else {"tile_indices": tile_indices}     # RFSamplerLong, :564-569
        if tile_indices is not None and not isinstance(self.sampler, S.RFSamplerLong):
            raise TypeError("tile_indices needs sampler_config.target = RFSamplerLong")
        if fused and isinstance(self.network, DiffusionTransformer):
            samples = self.sampler.sample_hip(self.network, randn, cond, uc, num_steps=num_steps, chunk_dim=chunk_dim, **extra)
        else:
            denoiser = lambda inp, sigma, c, **kw: self.denoiser(self.model, inp, sigma, c, concat_images=concat_images,
                                                                  chunk_dim=chunk_dim, **kw)
            samples = self.sampler(denoiser, randn, dict(cond), uc=dict(uc), num_steps=num_steps, **extra)
        samples = samples.to(self.dtype)
        if sp is not None:
            samples = sp.gather_to_rank0(samples, chunk_dim)
        return samples

    @torch.no_grad()
    def decode_first_stage(self, z):
        """diffusion_video.py:298-309."""
        if self.first_stage_model is None:
            raise RuntimeError("first stage (VAE) not built; pass build_first_stage: true in the model config")
        return self.first_stage_model.decode(1.0 / self.scale_factor * z)

    @torch.no_grad()
    def encode_first_stage(self, x, batch=None, force_encode=False):
        """diffusion_video.py:311-331."""
        if not force_encode and self.latent_input:
            return x * self.scale_factor
        if self.first_stage_model is None:
            raise RuntimeError("first stage (VAE) not built; pass build_first_stage: true in the model config")
        z = self.scale_factor * self.first_stage_model.encode(x)
        if self.sp is not None and self.sp.size > 1:
            self.sp.broadcast(z)
        return z

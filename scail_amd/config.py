"""Config plumbing: the reference builds every object with ``instantiate_from_config({target, params})``
(sgm/util.py:277-294) from OmegaConf-merged yaml (arguments.py:341-368).  OmegaConf is not available
offline, so this is a PyYAML equivalent that additionally maps the reference's hot-path ``target:``
strings onto the MI355X implementations -- the reference's own yaml files load unchanged."""
from __future__ import annotations

import importlib
from typing import Any, Dict

import yaml

# reference class path -> MI355X implementation (hot path only; everything else is left alone)
TARGET_MAP = {
    "dit_video_crossattn_sc_xc.DiffusionTransformer": "scail_amd.dit.DiffusionTransformer",
    "sgm.modules.diffusionmodules.denoiser.Denoiser": "scail_amd.sampler.Denoiser",
    "sgm.modules.diffusionmodules.denoiser_weighting.EpsWeighting": "scail_amd.sampler.EpsWeighting",
    "sgm.modules.diffusionmodules.denoiser_scaling.RFScaling": "scail_amd.sampler.RFScaling",
    "sgm.modules.diffusionmodules.sampling.RFSampler": "scail_amd.sampler.RFSampler",
    "sgm.modules.diffusionmodules.sampling.RFSamplerLong": "scail_amd.sampler.RFSamplerLong",
    "sgm.modules.diffusionmodules.discretizer.RFDiscretization": "scail_amd.sampler.RFDiscretization",
    "sgm.modules.diffusionmodules.guiders.VanillaCFG": "scail_amd.sampler.VanillaCFG",
    "sgm.modules.diffusionmodules.wrappers.OpenAIWrapper": "scail_amd.sampler.OpenAIWrapper",
    "sgm.models.wan_vae.WanVAE": "scail_amd.wan_vae.WanVAE",
    "sgm.modules.encoders.umt5.T5EncoderModel": "scail_amd.umt5.T5EncoderModel",   # strings (tokenizer_path) or ids + mask
    "sgm.modules.GeneralConditioner": "scail_amd.conditioner.GeneralConditioner",
    "sgm.modules.encoders.modules.GeneralConditioner": "scail_amd.conditioner.GeneralConditioner",
    "sgm.modules.encoders.clip.CLIPModel": "scail_amd.clip.CLIPModel",
}


def get_obj_from_str(string: str):
    string = TARGET_MAP.get(string, string)
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config: Dict[str, Any], **extra_kwargs):
    """sgm/util.py:277-294."""
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()), **extra_kwargs)


def load_yaml_configs(*paths: str) -> Dict[str, Any]:
    """Deep-merge yaml files left to right (OmegaConf.merge semantics for dicts)."""
    def merge(a, b):
        for k, v in b.items():
            if isinstance(v, dict) and isinstance(a.get(k), dict):
                merge(a[k], v)
            else:
                a[k] = v
        return a

    out: Dict[str, Any] = {}
    for p in paths:
        with open(p) as f:
            merge(out, yaml.safe_load(f) or {})
    return out

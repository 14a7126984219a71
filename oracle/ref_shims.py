"""Import the REAL reference (/root/reference) on CPU.  Build-container only.

TEST INFRASTRUCTURE ONLY (see oracle/scail_oracle.py header).  /root/reference does not
exist on the GPU box; nothing under tests/ -m gpu, smoke() or bench.py imports this.

Recipe (re-verified here):
  * MagicMock the packages sgm/__init__ pulls in but that are absent offline
    (pytorch_lightning, omegaconf, torchvision, beartype) -- after importing transformers;
  * init a 1-rank gloo group before any SAT model is built (sat/arguments.py:545-548
    would otherwise ask for device_id=cuda:0);
  * neutralise the two CUDA hard-wires: torch.cuda.get_device_name
    (sat/mpu/ulysses_attn_layer.py:36-37) and Tensor.cuda() (dit...:510-513);
  * run in fp32 (bf16-on-CPU crashes in sat/ops/layernorm.py:21-22).
"""
from __future__ import annotations

import argparse
import os
import sys
from unittest.mock import MagicMock

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "sat"))


_STATE = {}


def load_reference():
    """Returns the imported reference modules as a dict (idempotent)."""
    if _STATE:
        return _STATE
    if not available():
        raise RuntimeError("reference tree not present (this only works in the build container)")
    import torch
    import transformers  # noqa: F401  (must come before the torchvision stub)

    for m in ["pytorch_lightning", "omegaconf", "torchvision", "torchvision.utils",
              "torchvision.transforms", "beartype", "beartype.typing"]:
        if m not in sys.modules:
            sys.modules[m] = MagicMock()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:                 # single-process group: any free port (a fixed one collides between parallel test workers)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(so.getsockname()[1])
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=int(os.environ.get("RANK", 0)),
                                world_size=int(os.environ.get("WORLD_SIZE", 1)))
    torch.cuda.get_device_name = lambda *a, **k: "cpu"
    torch.Tensor.cuda = lambda self, *a, **k: self

    import dit_video_crossattn_sc_xc as dit
    from sgm.modules.diffusionmodules import denoiser, denoiser_scaling, denoiser_weighting  # noqa: F401
    from sgm.modules.diffusionmodules import discretizer, guiders, sampling, wrappers  # noqa: F401
    from sgm.models import wan_vae

    _STATE.update(dit=dit, sampling=sampling, denoiser=denoiser, wrappers=wrappers,
                  guiders=guiders, wan_vae=wan_vae, torch=torch)
    return _STATE


def build_reference_dit(cfg, state_dict):
    """Instantiate the reference DiffusionTransformer (dit...:1209-1321) for an oracle
    DiTConfig and load ``state_dict`` with strict=True (pins state_dict_spec)."""
    ref = load_reference()
    dit = ref["dit"]
    ta = argparse.Namespace(checkpoint_activations=False, vocab_size=1, max_sequence_length=64,
                            layernorm_order="pre", skip_init=False, model_parallel_size=1,
                            is_decoder=True)
    modules = {
        "pos_embed_config": {"target": "dit_video_crossattn_sc_xc.Rotary3DPositionEmbeddingMixin",
                             "params": {"hidden_size_head": cfg.head_dim, "interleaved_rope": True}},
        "patch_embed_config": {"target": "dit_video_crossattn_sc_xc.ImagePatchEmbeddingMixin",
                               "params": {"use_conv": True}},
        "adaln_layer_config": {"target": "dit_video_crossattn_sc_xc.AdaLNMixin",
                               "params": {"qk_ln": True, "qk_ln_affine": True,
                                          "hidden_size_head": cfg.hidden_size}},
        "final_layer_config": {"target": "dit_video_crossattn_sc_xc.FinalLayerMixin"},
    }
    net = dit.DiffusionTransformer(
        transformer_args=ta, num_frames=cfg.num_frames, time_compressed_rate=cfg.time_compressed_rate,
        latent_width=cfg.latent_width, latent_height=cfg.latent_height, patch_size=list(cfg.patch_size),
        in_channels=cfg.in_channels, out_channels=cfg.out_channels, hidden_size=cfg.hidden_size,
        text_dim=cfg.text_dim, num_layers=cfg.num_layers, num_attention_heads=cfg.num_attention_heads,
        elementwise_affine=False, time_freq_dim=cfg.time_freq_dim, time_embed_dim=cfg.time_embed_dim,
        share_adaln=True, inner_hidden_size=cfg.inner_hidden_size, use_SwiGLU=False, use_RMSNorm=False,
        layernorm_epsilon=cfg.layernorm_epsilon, modules=modules, dtype="fp32", use_i2v_clip=True)
    missing, unexpected = net.load_state_dict(state_dict, strict=True)
    assert not missing and not unexpected
    return net.eval()

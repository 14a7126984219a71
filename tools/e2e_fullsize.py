"""Full-size request through the driver (scail_amd.cli.run): SCAIL-14B shapes, 512x896x81f, random-init weights, VAE encode of
the reference frame and the half-resolution pose video, N sampler steps (default 2 instead of 50), VAE decode.  Checks shapes,
finiteness and value range; prints the stage times.  Everything below runs through the C-level executors."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import cli

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = {
    "model": {
        "use_i2v_clip": True, "scale_factor": 1.0, "build_first_stage": True,
        "network_config": {"target": "dit_video_crossattn_sc_xc.DiffusionTransformer", "params": dict(
            time_freq_dim=256, time_embed_dim=5120, share_adaln=True, elementwise_affine=False, num_frames=81,
            time_compressed_rate=4, latent_width=300, latent_height=300, num_layers=40, patch_size=[1, 2, 2], in_channels=20,
            out_channels=16, text_dim=4096, hidden_size=5120, inner_hidden_size=13824, num_attention_heads=40,
            transformer_args=dict(model_parallel_size=1, is_decoder=True),
            modules={"pos_embed_config": {"params": {"hidden_size_head": 128, "interleaved_rope": True}},
                     "adaln_layer_config": {"params": {"qk_ln": True, "hidden_size_head": 5120}}})},
        "first_stage_config": {"target": "sgm.models.wan_vae.WanVAE", "params": {"vae_pth": None, "dtype": "torch.bfloat16"}},
        "sampler_config": {"target": "sgm.modules.diffusionmodules.sampling.RFSampler", "params": dict(
            hunyuan_schedule=True, shift_scale=5, num_steps=50,
            guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})},
    },
    "args": {"sampling_image_size": [512, 896], "sampling_fps": 16},
}
t0 = time.perf_counter()
video, z, dt = cli.run(cfg, steps=steps, frames=81)
total = time.perf_counter() - t0
print(json.dumps(dict(case=f"SCAIL-14B shapes, 512x896x81f, {steps} sampler steps, random-init", latent=list(z.shape), video=list(video.shape),
                      finite=bool(torch.isfinite(video).all()), vmin=float(video.min()), vmax=float(video.max()),
                      request_s=dt, total_incl_init_s=total, peak_mem_GB=torch.cuda.max_memory_allocated() / 1e9)))

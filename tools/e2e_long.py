"""Long-video request at full size: RFSamplerLong temporal tiling (reference sampling.py:986-1085) over a 41-frame latent
(161 video frames, 512x896) with three overlapping 21-frame tiles, SCAIL-14B shapes, random-init weights, N sampler steps
(default 1), then VAE decode of the 161 frames.  Checks shapes and finiteness; prints times."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd.engine import SATVideoDiffusionEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda"
mc = {
    "use_i2v_clip": True, "scale_factor": 1.0, "build_first_stage": True,
    "network_config": {"target": "dit_video_crossattn_sc_xc.DiffusionTransformer", "params": dict(
        time_freq_dim=256, time_embed_dim=5120, share_adaln=True, elementwise_affine=False, num_frames=81,
        time_compressed_rate=4, latent_width=300, latent_height=300, num_layers=40, patch_size=[1, 2, 2], in_channels=20,
        out_channels=16, text_dim=4096, hidden_size=5120, inner_hidden_size=13824, num_attention_heads=40,
        transformer_args=dict(model_parallel_size=1, is_decoder=True),
        modules={"pos_embed_config": {"params": {"hidden_size_head": 128, "interleaved_rope": True}},
                 "adaln_layer_config": {"params": {"qk_ln": True, "hidden_size_head": 5120}}})},
    "first_stage_config": {"target": "sgm.models.wan_vae.WanVAE", "params": {"vae_pth": None, "dtype": "torch.bfloat16"}},
    "sampler_config": {"target": "sgm.modules.diffusionmodules.sampling.RFSamplerLong", "params": dict(
        hunyuan_schedule=True, shift_scale=5, num_steps=50,
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})},
}
eng = SATVideoDiffusionEngine(mc, device=dev)
T, H, W, Tt = 41, 64, 112, 21
tiles = [list(range(s, s + Tt)) for s in (0, 10, 20)]
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g)
ctx = r(1, 512, 4096); ctx[:, 64:] = 0
uctx = torch.zeros(1, 512, 4096); uctx[:, :1] = r(1, 1, 4096)
shared = dict(concat_images=torch.zeros(1, device=dev), ref_concat=r(1, 1, 16, H, W).to(dev).to(torch.bfloat16),
              smpl_tiled=r(1, len(tiles), Tt, 16, H // 2, W // 2).to(dev).to(torch.bfloat16),
              image_clip_features=r(1, 257, 1280).to(dev).to(torch.bfloat16))
c = dict(crossattn=ctx.to(dev), **shared)
uc = dict(crossattn=uctx.to(dev), **shared)
torch.cuda.synchronize(); t0 = time.perf_counter()
z = eng.sample(c, uc=uc, batch_size=1, shape=(T, 16, H, W), num_steps=steps, tile_indices=tiles, generator=torch.Generator().manual_seed(1))
torch.cuda.synchronize(); t1 = time.perf_counter()
x = eng.decode_first_stage(z.permute(0, 2, 1, 3, 4).contiguous().float())
torch.cuda.synchronize(); t2 = time.perf_counter()
print(json.dumps(dict(case=f"long video: 41-frame latent, 3 tiles of 21, {steps} step(s)", latent=list(z.shape), video=list(x.shape),
                      finite=bool(torch.isfinite(x).all()), sample_s=t1 - t0, decode_s=t2 - t1,
                      peak_mem_GB=torch.cuda.max_memory_allocated() / 1e9)))

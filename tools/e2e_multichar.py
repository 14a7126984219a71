"""BASELINE config 5 (multi-character in-context concat; an extension, not in the reference) at FULL SIZE on one GPU:
SCAIL-14B shapes, 512x896x81f latent, 2 reference frames + 2 pose streams -> L = 3584 + 37632 + 18816 = 60032 tokens.
Times sampler steps (batch-2 CFG forward + Euler) through the per-op host path; perf-only, parity is covered at tiny size
(tests/test_dit_gpu.py::test_multi_character_extension_vs_oracle).
usage: e2e_multichar.py [steps] [n_char]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import ops
from scail_amd.dit import DiffusionTransformer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
C = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = "cuda"
net = DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=81, latent_width=300, latent_height=300,
                           hidden_size=5120, num_layers=40, num_attention_heads=40, inner_hidden_size=13824, text_dim=4096,
                           time_freq_dim=256, time_embed_dim=5120, share_adaln=True, use_i2v_clip=True, device=dev, init_seed=1234)
g = torch.Generator().manual_seed(1)
T, H, W = 21, 64, 112
x = torch.randn(1, T, 16, H, W, generator=g).to(dev)
ref = torch.randn(1, C, 16, H, W, generator=g).to(dev).to(torch.bfloat16)
pose = torch.randn(1, C * T, 16, H // 2, W // 2, generator=g).to(dev).to(torch.bfloat16)
ctx = torch.randn(2, 512, 4096, generator=g).to(dev).to(torch.bfloat16)
clip = torch.randn(1, 257, 1280, generator=g).to(dev).to(torch.bfloat16)
kw = dict(concat_images=torch.zeros(1, device=dev), image_clip_features=clip, ref_concat=ref, concat_smpl_render=pose)
sig = torch.linspace(1.0, 0.9, steps + 2)


def step(i):
    t = (sig[i] * 1000.0).repeat(2).to(dev)
    v = net.forward_f32(torch.cat([x, x], 0), t, ctx, None, cond_key="mc", **kw)
    ops.cfg_euler_(x, v, 4.0, float(sig[i + 1] - sig[i]))


step(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1, steps + 1):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
L = C * (H // 2) * (W // 2) + T * (H // 2) * (W // 2) + C * T * (H // 4) * (W // 4)
print(json.dumps(dict(case=f"multi-character extension, {C} ref + {C} pose streams, full size", tokens=L, s_per_step=dt,
                      latent_tokens_per_s=T * (H // 2) * (W // 2) / dt, finite=bool(torch.isfinite(x).all()),
                      x_abs_mean=float(x.abs().mean()), peak_mem_GB=torch.cuda.max_memory_allocated() / 1e9)))

#!/usr/bin/env python
"""Experiment: the two CFG elements of a sampler step are independent network evaluations.  Does running them as TWO concurrent B = 1
executor calls on two streams (memory-bound row passes of one element under the MFMA-bound kernels of the other; partial last rounds
filled) beat the ONE B = 2 call of the product path?  Two network objects with the same seed (= the same weights, twice in memory) so
that each call has its own workspace.  Prints s per step for both forms and the largest difference of the results."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd.dit import DiffusionTransformer  # noqa: E402

dev = "cuda"


def make():
    n = DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=81, latent_width=300, latent_height=300,
                             hidden_size=5120, num_layers=40, num_attention_heads=40, inner_hidden_size=13824, text_dim=4096,
                             time_freq_dim=256, time_embed_dim=5120, share_adaln=True, use_i2v_clip=True, device=dev, init_seed=1234)
    n.cache_conditioning = False
    return n


nets = [make(), make()]
g = torch.Generator().manual_seed(1)
T, H, W = 21, 64, 112
ctx = torch.randn(2, 512, 4096, generator=g).to(dev).to(torch.bfloat16)
clip = torch.randn(1, 257, 1280, generator=g).to(dev).to(torch.bfloat16)
x = torch.randn(1, T, 16, H, W, generator=g).to(dev)
ref = torch.randn(1, 1, 16, H, W, generator=g).to(dev).to(torch.bfloat16)
pose = torch.randn(1, T, 16, H // 2, W // 2, generator=g).to(dev).to(torch.bfloat16)
kw = dict(concat_images=torch.zeros(1, device=dev), image_clip_features=clip, ref_concat=ref, concat_smpl_render=pose, chunk_dim=None)
tt = torch.tensor([700.0, 700.0], device=dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def joint():
    return nets[0].forward_f32(torch.cat([x, x], 0), tt, ctx, None, **kw)


def dual():
    main = torch.cuda.current_stream()
    outs = []
    for b in range(2):
        streams[b].wait_stream(main)
        with torch.cuda.stream(streams[b]):
            outs.append(nets[b].forward_f32(x, tt[b:b + 1], ctx[b:b + 1], None, **kw))
    for s in streams:
        main.wait_stream(s)
    return torch.cat(outs, 0)


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


a = joint(); b = dual()
torch.cuda.synchronize()
rec = {"max_abs_diff_joint_vs_dual": float((a - b).abs().max())}
for rnd in range(2):
    rec.setdefault("joint_B2_s", []).append(round(timed(joint), 4))
    rec.setdefault("dual_stream_B1x2_s", []).append(round(timed(dual), 4))
print(json.dumps(rec))

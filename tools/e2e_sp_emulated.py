"""Sequence-parallel data path at FULL SIZE on one GPU: N virtual ranks (threads, scail_amd.parallel.ThreadBackend), each with
its own SCAIL-14B-shaped network built from the same seed, run ONE network evaluation on their H-slab of the 512x896x81f latent
(rank-shifted RoPE, per-layer exchange in the chosen mode, gather to rank 0); compared with the single-rank evaluation.
usage: e2e_sp_emulated.py <world> <allgather|ulysses> [layers] [n_char]   (layers < 40 keeps the run short; shapes stay full
size; n_char = 2 is the multi-character extension of BASELINE config 5: 60 032 tokens)"""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd.dit import DiffusionTransformer
from scail_amd.parallel import SequenceParallel, ThreadBackend

world, mode = int(sys.argv[1]), sys.argv[2]
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
C = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = "cuda"
P = dict(hidden_size=5120, num_layers=layers, num_attention_heads=40, inner_hidden_size=13824, text_dim=4096,
         time_freq_dim=256, time_embed_dim=5120)
mk = lambda: DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=81, latent_width=300, latent_height=300,
                                  share_adaln=True, use_i2v_clip=True, device=dev, init_seed=1234, **P)
g = torch.Generator().manual_seed(1)
T, H, W = 21, 64, 112
x = torch.randn(2, T, 16, H, W, generator=g).to(dev)
ref = torch.randn(1, C, 16, H, W, generator=g).to(dev).to(torch.bfloat16)
pose = torch.randn(1, C * T, 16, H // 2, W // 2, generator=g).to(dev).to(torch.bfloat16)
ctx = torch.randn(2, 512, 4096, generator=g).to(dev).to(torch.bfloat16)
clip = torch.randn(1, 257, 1280, generator=g).to(dev).to(torch.bfloat16)
t = torch.tensor([700.0, 700.0], device=dev)
kw = dict(concat_images=torch.zeros(1, device=dev), image_clip_features=clip)
net = mk()
single = net.forward_f32(x, t, ctx, None, ref_concat=ref, concat_smpl_render=pose, **kw)
torch.cuda.synchronize()
del net
shared = ThreadBackend.Shared(world)
outs, errs = [None] * world, []


def run(rk):
    try:
        torch.cuda.set_device(0)
        n = mk()
        sp = SequenceParallel(ThreadBackend(shared, rk), mode=mode)
        n.sp = sp
        sp.check_latent(H, W, 3)
        ch = lambda tt: sp.chunk(tt, 3)
        o = n.forward_f32(ch(x), t, ctx, None, ref_concat=ch(ref), concat_smpl_render=ch(pose), chunk_dim=3, **kw)
        outs[rk] = sp.gather_to_rank0(o, 3)
    except Exception as e:  # pragma: no cover
        errs.append(repr(e))
        shared.barrier.abort()


t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(rk,)) for rk in range(world)]
[tt.start() for tt in th]
[tt.join() for tt in th]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if errs:
    print(json.dumps(dict(case="sp emulated", world=world, mode=mode, errors=errs)))
    sys.exit(1)
d = (outs[0] - single).abs()
print(json.dumps(dict(case=f"SP emulated at full size: {world} virtual ranks, {mode}, {layers} layers, {C} character(s)", max_abs_diff=float(d.max()),
                      mean_abs_diff=float(d.mean()), ref_abs_mean=float(single.abs().mean()), seconds=dt,
                      peak_mem_GB=torch.cuda.max_memory_allocated() / 1e9)))

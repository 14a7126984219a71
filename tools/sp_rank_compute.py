"""Compute-only time of ONE sequence-parallel rank's share of the config-2 sampler step, on one GPU: the collectives are
replaced by local copies (results are meaningless, the launch sequence and every kernel shape are those of rank 0 in an
N-rank run).  Gives the compute part of the strong-scaling efficiency -- (T_1 / N) / T_N -- without an N-GPU node; the
exchange time comes on top.
The timed path is the product's: one scail_dit_step / scail_dit_step_sp call of the C executor per network evaluation, the exchange
callback served by local copies.  `--host` times the per-op host path (scail_amd.parallel, ~30 binding calls per layer) beside it.
`--comm-wgs=K[,K2..]`: serve the exchanges with scail_amd.parallel.ConcurrentCopyBackend instead -- a K-workgroup kernel on a side stream
that moves the message and stays resident for the xGMI transfer time (`--link-gbps=48` per direction and peer link), i.e. the CU occupancy
of RCCL's channels beside the attention; `--plan-cus=1` also tells the attention launch plan about it (option "attn4_cus" = CUs - K).
usage: sp_rank_compute.py [--host] [--no-pair] [--rows=256|192|0] [--xcd=0|1] [--comm-wgs=K,..] [--plan-cus=0|1] [--link-gbps=G] [N ...]      e.g. 1 2 4 8"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import ops
from scail_amd.dit import DiffusionTransformer
from scail_amd.parallel import ConcurrentCopyBackend, LocalCopyBackend, SequenceParallel


dev = "cuda"
net = DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=81, latent_width=300, latent_height=300,
                           hidden_size=5120, num_layers=40, num_attention_heads=40, inner_hidden_size=13824, text_dim=4096,
                           time_freq_dim=256, time_embed_dim=5120, share_adaln=True, use_i2v_clip=True, device=dev, init_seed=1234)
net.cache_conditioning = False
g = torch.Generator().manual_seed(1)
T, H, W = 21, 64, 112
ctx = torch.randn(2, 512, 4096, generator=g).to(dev).to(torch.bfloat16)
clip = torch.randn(1, 257, 1280, generator=g).to(dev).to(torch.bfloat16)
base = None
HOST = "--host" in sys.argv
PAIR = "--no-pair" not in sys.argv        # SCAIL_DIT_CFG_PAIR, the sampler's default (layer 0 up to the first cross attention evaluated once)
from scail_amd import lib
for a in sys.argv[1:]:                    # same-process A/B of the attention launch shape: --rows=256|192|0 (0 = per-launch choice), --xcd=0|1
    if a.startswith("--rows="):
        lib.set_option("attn4_rows", int(a[7:]))
    if a.startswith("--xcd="):
        lib.set_option("attn4_xcd", int(a[6:]))
COMM = [None]
LINK, PLAN_CUS = 48.0, False
for a in sys.argv[1:]:
    if a.startswith("--comm-wgs="):
        COMM = [None] + [int(v) for v in a[11:].split(",")]
    if a.startswith("--link-gbps="):
        LINK = float(a[12:])
    if a.startswith("--plan-cus="):
        PLAN_CUS = a[11:] == "1"
CUS = torch.cuda.get_device_properties(0).multi_processor_count
for N, K in [(int(a), k) for a in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["1", "2", "4", "8"]) for k in (COMM if int(a) > 1 else [None])]:
    sp = SequenceParallel(LocalCopyBackend(N) if K is None else ConcurrentCopyBackend(N, K, LINK)) if N > 1 else None
    lib.set_option("attn4_cus", CUS - K if (K is not None and PLAN_CUS) else 0)
    net.sp = sp
    h = H // N
    x = torch.randn(1, T, 16, h, W, generator=g).to(dev)
    ref = torch.randn(1, 1, 16, h, W, generator=g).to(dev).to(torch.bfloat16)
    pose = torch.randn(1, T, 16, h // 2, W // 2, generator=g).to(dev).to(torch.bfloat16)
    kw = dict(concat_images=torch.zeros(1, device=dev), image_clip_features=clip, ref_concat=ref, concat_smpl_render=pose,
              chunk_dim=3 if N > 1 else None)

    def step():
        v = net.forward_f32(torch.cat([x, x], 0), torch.tensor([700.0, 700.0], device=dev), ctx, None, cfg_pair=PAIR, **kw)
        ops.cfg_euler_(x, v, 4.0, -0.01)

    def timed():
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 2

    net.use_c_step = True
    dt = timed()
    base = base or dt * N
    mode = sp.resolve_mode(40) if sp else "-"
    rows = lib.load().scail_flash_attn_rows_for(1 if sp else 2, 40 // N if mode == "ulysses" else 40, 48832 if mode in ("-", "ulysses") else 48832 // N)
    rec = dict(ranks=N, mode=mode, path="C executor", s_per_step_one_rank=dt, compute_only_efficiency=(base / N) / dt, attn_query_tile_rows=rows, cfg_pair=PAIR)
    if K is not None:
        rec.update(comm_standin_workgroups=K, link_gbps=LINK, plan_knows_cus=PLAN_CUS,
                   note="collectives = a K-workgroup resident kernel on a side stream (message copy + xGMI transfer time); efficiency includes what is exposed of it")
    if HOST and N > 1:
        net.use_c_step = False
        rec["s_per_step_one_rank_host_path"] = timed()
        rec["compute_only_efficiency_host_path"] = (base / N) / rec["s_per_step_one_rank_host_path"]
    print(json.dumps(rec), flush=True)

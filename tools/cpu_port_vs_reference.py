"""Does the CPU port cost what the reference costs?    bench.py's `cpu_baseline` times the ORACLE's block
(oracle/scail_oracle.py O.block) because the reference cannot travel to the GPU box; this script, run in the BUILD container where both
are importable, times the reference's own `AdaLNMixin.layer_forward` (dit_video_crossattn_sc_xc.py:1009-1051, inside the real
`DiffusionTransformer.forward`) and `O.block` (inside `O.dit_forward`) on the same inputs, weights and thread count at the real width
(D = 5120, 40 heads, FF = 13 824, B = 2) and the two shortest lengths of the baseline's per-token fit (L = 1008, 2128).  The ratio
port / reference goes into BASELINE.md section 3 and into the bench line as `cpu_baseline.port_vs_reference_time_ratio` (a constant with
this provenance).  Build container only (reads /root/reference).
usage: python tools/cpu_port_vs_reference.py [threads]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shims
from oracle import scail_oracle as O

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
torch.set_num_threads(threads)
cfg = O.DiTConfig(hidden_size=5120, num_layers=1, num_attention_heads=40, inner_hidden_size=13824, text_dim=64, time_freq_dim=256,
                  time_embed_dim=5120, latent_height=64, latent_width=64, num_frames=13)
sd = O.make_state_dict(cfg, seed=3)
net = ref_shims.build_reference_dit(cfg, sd)
mix = net.mixins["adaln_layer"]
orig = mix.layer_forward
ref_t = [0.0]


def timed_layer(*a, **k):
    t0 = time.perf_counter()
    o = orig(*a, **k)
    ref_t[0] += time.perf_counter() - t0
    return o


mix.layer_forward = timed_layer
net.collect_hooks_()
port_t = [0.0]
oblock = O.block


def timed_block(*a, **k):
    t0 = time.perf_counter()
    o = oblock(*a, **k)
    port_t[0] += time.perf_counter() - t0
    return o


O.block = timed_block
g = torch.Generator().manual_seed(1)
res = []
for T in (1, 3):
    x = torch.randn(2, T, 16, 32, 56, generator=g)
    ref = torch.randn(1, 1, 16, 32, 56, generator=g)
    pose = torch.randn(1, T, 16, 16, 28, generator=g)
    ctx = torch.randn(2, 512, 64, generator=g)
    clip = torch.randn(1, 257, 1280, generator=g)
    t = torch.tensor([700.0, 700.0])
    L = (1 + T) * 16 * 28 + T * 8 * 14
    best = {"reference": 1e9, "port": 1e9}
    outs = {}
    for rep in range(5):                     # first pass = warm-up of both; best of the next four, interleaved
        ref_t[0] = port_t[0] = 0.0
        with torch.no_grad():
            outs["reference"] = net(x, timesteps=t, context=ctx, concat_images=torch.zeros(1, *x.shape[1:]), ref_concat=ref,
                                    concat_smpl_render=pose, image_clip_features=clip)
            outs["port"] = O.dit_forward(cfg, sd, x, t, ctx, ref, pose, clip)
        if rep:
            best["reference"] = min(best["reference"], ref_t[0])
            best["port"] = min(best["port"], port_t[0])
    err = float((outs["reference"] - outs["port"]).abs().max())
    res.append(dict(L=L, reference_layer_forward_s=best["reference"], port_block_s=best["port"], port_over_reference=best["port"] / best["reference"],
                    max_abs_diff_of_the_network_outputs=err))
    print(json.dumps(res[-1]), flush=True)
ratio = sum(r["port_block_s"] for r in res) / sum(r["reference_layer_forward_s"] for r in res)
print(json.dumps(dict(threads=threads, hardware_threads=os.cpu_count(), width="D=5120, 40 heads, FF=13824, B=2", port_vs_reference_time_ratio=ratio,
                      torch=torch.__version__)))

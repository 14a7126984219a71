"""Data sensitivity of the self-attention kernel's measured rate: scail_attn4_m16f runs an OPTIMISTIC pass and a
workgroup in which a score exceeds the first key tile's row maximum by more than ~167 log2 units runs again with the lazy-maximum loop.
Random data never does; trained q / k norm weights may.  This probe times the config-2 launch (B = 2, 40 heads, L = 48 832) with a CHOSEN
fraction of workgroups forced to restart -- one query row of the workgroup is aligned with one key outside the first tile, 12x its norm
(~196 log2 units above the row's first-tile maximum) -- and counts the restarts with the kernel's own counter
(scail_flash_attn_count_restarts).  Same process, same tensors otherwise.
usage: attn_restart_probe.py [fraction ...]      default 0 0.001 0.01 0.1"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib as L, ops

dev = "cuda"
B, H, Lq = 2, 40, 48832
D = H * 128
g = torch.Generator(device=dev).manual_seed(3)
q = torch.randn(B, Lq, D, device=dev, generator=g)
k0 = torch.randn(B, Lq, D, device=dev, generator=g).to(torch.bfloat16)
v = torch.randn(B, Lq, D, device=dev, generator=g).to(torch.bfloat16)
vt = ops.transpose_v(v, H)
del v
qb = (q * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
rows = int(L.load().scail_flash_attn_rows_for(B, H, Lq))
assert rows == 256, rows
n_tiles = (Lq + 255) // 256
W = B * H * n_tiles
ctr = torch.zeros(1, device=dev, dtype=torch.int32)
flop = 4.0 * Lq * Lq * 128 * H * B
fracs = [float(a) for a in sys.argv[1:]] or [0.0, 0.001, 0.01, 0.1]
base_ms = None
for f in fracs:
    k = k0.clone()
    n = int(round(f * W))
    gen = torch.Generator().manual_seed(17)
    pick = torch.randperm(W, generator=gen)[:n]
    for w in pick.tolist():
        pair, tile = divmod(w, n_tiles)
        b, h = divmod(pair, H)
        r = min(tile * 256 + int(torch.randint(0, 256, (1,), generator=gen)), Lq - 1)
        j = int(torch.randint(64, Lq, (1,), generator=gen))
        k[b, j, h * 128:(h + 1) * 128] = (q[b, r, h * 128:(h + 1) * 128] * 12.0).to(torch.bfloat16)
    out = torch.empty(B, Lq, D, device=dev, dtype=torch.bfloat16)
    ops.flash_attn(qb, k, vt, out=out, q_prescaled=True)                 # warm-up
    ctr.zero_()
    L.call("scail_flash_attn_count_restarts", ctr.data_ptr())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 4
    ev[0].record()
    for _ in range(reps):
        ops.flash_attn(qb, k, vt, out=out, q_prescaled=True)
    ev[1].record()
    torch.cuda.synchronize()
    L.call("scail_flash_attn_count_restarts", None)
    ms = ev[0].elapsed_time(ev[1]) / reps
    base_ms = base_ms or ms
    rec = dict(workgroups=W, forced_fraction=f, forced=n, restarts_per_launch=int(ctr.item()) / reps, ms_per_launch=ms, tflops=flop / ms / 1e9,
               slowdown_vs_first=ms / base_ms - 1.0, finite=bool(torch.isfinite(out.float()).all()))
    print(json.dumps(rec), flush=True)
    del k, out

#!/usr/bin/env python
"""Phase budget of scail_attn4_x2 (option cross4 = 1) at the config-2 query shape (B = 2, 40 heads, 48 832 queries): the launch timed over
key-set sizes that separate the per-item fixed cost (1 + 1 tiles), the per-tile cost in the remainder chain (sets below 10 tiles never
reach the hot loop) and in the hot loop (long sets), beside cross_attn2_kernel on the same shapes.  One JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib, ops  # noqa: E402

DEV = "cuda"
B, H, Lq = 2, 40, 48832
D = H * 128
ITEMS = B * H * ((Lq + 255) // 256)
ROUNDS = ITEMS / 256.0


def timeit(fn, iters=7):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)
q = (torch.randn(B, Lq, D, device=DEV, generator=g) * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
o = torch.empty(B, Lq, D, device=DEV, dtype=torch.bfloat16)
for Lk1, Lk2 in [(64, 64), (128, 64), (256, 64), (512, 64), (512, 257), (512, 320), (768, 64), (1024, 64), (2048, 64), (4096, 64), (1024, 1024)]:
    k1, v1, k2, v2 = rn(B, Lk1, D), rn(B, Lk1, D), rn(1, Lk2, D), rn(1, Lk2, D)
    vt1, vt2 = ops.transpose_v(v1, H), ops.transpose_v(v2, H)
    rec = {"Lk1": Lk1, "Lk2": Lk2, "tiles": (Lk1 + 63) // 64 + (Lk2 + 63) // 64}
    for name, c4 in (("x2", 1), ("hipcc", 0)):
        lib.set_option("cross4", c4)
        ms = timeit(lambda: ops.cross_attn2(q, k1, vt1, k2, vt2, out=o, q_prescaled=True))
        rec[name + "_ms"] = ms
        rec[name + "_us_per_item"] = ms * 1e3 / ROUNDS
    lib.set_option("cross4", 2)
    print(json.dumps(rec), flush=True)

#!/usr/bin/env python
"""Time the text + CLIP cross attention of one DiT layer at the config-2 shape: the fused two-key-set launch
(scail_cross_attn2_bf16) against the two launches it replaced (flash_attn + flash_attn accumulate).  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import ops  # noqa: E402

DEV = "cuda"
B, H, Lq, Lt, Lc = 2, 40, 48832, 512, 257
D = H * 128


def timeit(fn, iters=10):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)
qkv = rn(B, Lq, 3 * D)
q = qkv[..., :D]
k1, v1, k2, v2 = rn(B, Lt, D), rn(B, Lt, D), rn(1, Lc, D), rn(1, Lc, D)
vt1, vt2 = ops.transpose_v(v1, H), ops.transpose_v(v2, H)
o = torch.empty(B, Lq, D, device=DEV, dtype=torch.bfloat16)
o2 = torch.empty_like(o)


def two():
    ops.flash_attn(q, k1, vt1, out=o2)
    ops.flash_attn(q, k2, vt2, out=o2, accumulate=True)


ms1 = timeit(lambda: ops.cross_attn2(q, k1, vt1, k2, vt2, out=o))
ms2 = timeit(two)
fl = 4.0 * B * H * Lq * (Lt + Lc) * 128
print(json.dumps({"shape": {"B": B, "heads": H, "Lq": Lq, "Lt": Lt, "Lc": Lc}, "fused_ms": ms1, "two_launch_ms": ms2,
                  "fused_TFLOPs": fl / ms1 / 1e9, "two_launch_TFLOPs": fl / ms2 / 1e9,
                  "max_abs_diff": float((o.float() - o2.float()).abs().max())}))

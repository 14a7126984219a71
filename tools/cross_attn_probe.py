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


from scail_amd import lib  # noqa: E402
fl = 4.0 * B * H * Lq * (Lt + Lc) * 128
qs = (q.float() * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
res = {"shape": {"B": B, "heads": H, "Lq": Lq, "Lt": Lt, "Lc": Lc}}
outs = {}
for name, c4 in (("attn4_x2", 1), ("cross_attn2_kernel", 0)):        # same process: the generated persistent kernel, the round-2 hipcc kernel
    lib.set_option("cross4", c4)
    ms = timeit(lambda: ops.cross_attn2(qs, k1, vt1, k2, vt2, out=o, q_prescaled=True))
    outs[name] = o.clone()
    res[name] = {"ms": ms, "TFLOPs": fl / ms / 1e9, "frac_of_2500": fl / ms / 1e9 / 2500.0}
    if c4:
        ms = timeit(lambda: ops.cross_attn2(q, k1, vt1, k2, vt2, out=o))
        res[name]["ms_raw_scale_q"] = ms
lib.set_option("cross4", 2)
ms2 = timeit(two)
res["two_launch_ms"] = ms2
res["max_abs_diff_x2_vs_hipcc"] = float((outs["attn4_x2"].float() - outs["cross_attn2_kernel"].float()).abs().max())
print(json.dumps(res))

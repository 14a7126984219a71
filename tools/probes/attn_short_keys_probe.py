#!/usr/bin/env python
"""How does scail_attn4_m16f behave on SHORT key sets (the cross-attention regime: 512 text + 257 CLIP keys against 48 832 queries)?
Times scail_flash_attn_bf16 at Lq = 48 832, B = 2, 40 heads for Lk in {512, 768, 1024, 2048, 4096} (attn4 path) and the fused two-set
hipcc kernel (scail_cross_attn2_bf16) beside it: the per-launch fixed cost of attn4's prologue / epilogue = the intercept of ms(Lk)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import ops, lib as L  # noqa: E402

DEV = "cuda"
B, H, Lq = 2, 40, 48832
D = H * 128


def timeit(fn, iters=8):
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
o = torch.empty(B, Lq, D, device=DEV, dtype=torch.bfloat16)
out = {}
MODES = ((1, "attn4"), (0, "8wave_swp")) if "--both" in sys.argv else ((1, "attn4"),)
for on, tag in MODES:
  L.set_option("attn4", on)
  for Lk in (256, 320, 512, 768, 832, 1024, 2048, 4096):
    if on and Lk < 512:
        continue
    k, v = rn(B, Lk, D), rn(B, Lk, D)
    vt = ops.transpose_v(v, H)
    kind = L.load().scail_flash_attn_kernel_for(q.stride(1), k.stride(1), o.stride(1), Lq, Lk, 0, 0)
    ms = timeit(lambda: ops.flash_attn(q, k, vt, out=o))
    out[f"{tag}_Lk{Lk}"] = {"kernel": kind, "ms": round(ms, 4), "TFLOPs": round(4.0 * B * H * Lq * Lk * 128 / ms / 1e9, 1)}
L.set_option("attn4", 1)
k1, v1, k2, v2 = rn(B, 512, D), rn(B, 512, D), rn(1, 257, D), rn(1, 257, D)
vt1, vt2 = ops.transpose_v(v1, H), ops.transpose_v(v2, H)
ms = timeit(lambda: ops.cross_attn2(q, k1, vt1, k2, vt2, out=o))
out["cross_attn2_512+257"] = {"ms": round(ms, 4), "TFLOPs": round(4.0 * B * H * Lq * 769 * 128 / ms / 1e9, 1)}
print(json.dumps(out))

#!/usr/bin/env python
"""Same-process A/B of the row passes of a DiT layer at config 2 (M = 2 x 48 832 rows, D = 5120): block-per-row kernels (option
row_wave = 0) against the one-wave-per-row kernels (row_wave = 1), as the executor launches them: LayerNorm + modulate (dense in / out),
q and k RMSNorm + RoPE IN PLACE on the (B, L, 3 D) qkv buffer (row stride 3 D), the affine LayerNorm, transpose_v.  Prints ms, achieved
GB/s on the algorithmic bytes (x read + y written) and the largest difference between the two forms (order of the fp32 sums only)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import ops, lib as L  # noqa: E402

DEV = "cuda"
B, Ltok, H = 2, 48832, 40
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
x = torch.randn(B, Ltok, D, device=DEV, generator=g).to(torch.bfloat16)
qkv = torch.randn(B, Ltok, 3 * D, device=DEV, generator=g).to(torch.bfloat16)
shift, scale = torch.randn(B, D, device=DEV, generator=g) * 0.1, torch.randn(B, D, device=DEV, generator=g) * 0.1
w, bb = 1 + 0.1 * torch.randn(D, device=DEV, generator=g), 0.1 * torch.randn(D, device=DEV, generator=g)
ang = torch.rand(Ltok, 64, device=DEV, generator=g) * 6.28
cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
y = torch.empty_like(x)
qk_out = torch.empty(B, Ltok, D, device=DEV, dtype=torch.bfloat16)
cases = {
    "ln_modulate": lambda: ops.ln_modulate(x, shift, scale, out=y),
    "layernorm_affine": lambda: ops.layernorm_affine(x, w, bb, out=y),
    "rmsnorm_rope q (strided in, dense out)": lambda: ops.rmsnorm_rope(qkv[..., :D], w, cos, sin, out=qk_out, rows_per_batch=Ltok, out_scale=0.1275),
    "rmsnorm_rope k IN PLACE on the qkv buffer (timing only)": lambda: ops.rmsnorm_rope(qkv[..., D:2 * D], w, cos, sin, rows_per_batch=Ltok),
    "rmsnorm (no rope, dense)": lambda: ops.rmsnorm_rope(x, w, out=y, rows_per_batch=Ltok),
}
alg = 2 * B * Ltok * D * 2
res = {}
for name, fn in cases.items():
    rec = {}
    outs = []
    for rnd in range(2):
        for mode in (0, 1):
            L.set_option("row_wave", mode)
            ms = timeit(fn)
            rec.setdefault(f"row_wave{mode}_ms", []).append(round(ms, 4))
            if rnd == 0:
                outs.append((qk_out if "strided" in name else y).clone())
    rec["GBps"] = {m: round(alg / min(rec[f"row_wave{m}_ms"]) / 1e6, 0) for m in (0, 1)}
    rec["max_abs_diff"] = float((outs[0].float() - outs[1].float()).abs().max())
    rec["n_diff"] = int((outs[0] != outs[1]).sum())
    res[name] = rec
L.set_option("row_wave", 1)
v = qkv[..., 2 * D:]
ms = timeit(lambda: ops.transpose_v(v, H))
res["transpose_v"] = {"ms": round(ms, 4), "GBps": round(alg / ms / 1e6, 0)}
print(json.dumps(res, indent=1))

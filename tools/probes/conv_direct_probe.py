#!/usr/bin/env python
"""Same-process A/B of the VAE's HBM-bound convolutions at BASELINE config 4's sizes: the direct-gather kernel (option conv_direct = 1) against
the gather kernel (0).  One JSON line per shape with ms and GB/s on the algorithmic bytes (input read once + output written once)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import lib as L, ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=6):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


g = torch.Generator(device=DEV).manual_seed(0)
cases = [("stem 3->96 3x3x3", (81, 512, 896, 8), 96, (3, 3, 3), (1, 1, 1), None),
         ("shortcut 96->192 1x1x1", (81, 256, 448, 96), 192, (1, 1, 1), (1, 1, 1), None),
         ("shortcut 192->384 1x1x1", (41, 128, 224, 192), 384, (1, 1, 1), (1, 1, 1), None),
         ("shortcut 192->384 1x1x1 (decoder)", (41, 128, 224, 192), 384, (1, 1, 1), (1, 1, 1), None)]
for name, xs, cout, k, stride, pad in cases[:3]:
    T, H, W, cin = xs
    x = torch.randn(T, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    w = torch.randn(cout, cin, *k, device=DEV, generator=g) / (cin * k[0] * k[1] * k[2]) ** 0.5
    wp = ops.prep_conv_weight(w, torch.randn(cout, device=DEV, generator=g), cin_pad=cin)
    To = T if pad is None else (T - k[0]) // stride[0] + 1
    kw = {} if pad is None else dict(stride=stride, pad=pad)
    y = torch.empty(To, H, W, cout, device=DEV, dtype=torch.bfloat16)
    alg = (x.numel() + y.numel()) * 2
    rec = {"conv": name, "in": list(xs), "algorithmic_GB": round(alg / 1e9, 2)}
    outs = {}
    for rnd in range(2):
        for mode, nm in ((0, "gather"), (1, "direct")):
            L.set_option("conv_direct", mode)
            rec.setdefault(nm + "_ms", []).append(round(timeit(lambda: ops.conv3d_cl(x, wp, (To, H, W), out=y, **kw)), 3))
            outs[nm] = y.clone()
    L.set_option("conv_direct", 1)
    for nm in ("gather", "direct"):
        rec[nm + "_GBps"] = round(alg / min(rec[nm + "_ms"]) / 1e6, 0)
    rec["max_abs_diff"] = float((outs["gather"].float() - outs["direct"].float()).abs().max())
    print(json.dumps(rec), flush=True)
    del x, y, outs

#!/usr/bin/env python
"""(round 6) Same-process A/B of Resample's stride-2 downsampling convolutions at BASELINE config 4's sizes: conv_s2_kernel with one / two output
frames per workgroup (option conv_s2 = 1 / 2) against the gather kernel (0).  One JSON line per shape: ms, TFLOP/s, GB/s on the algorithmic bytes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib as L, ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=8):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


g = torch.Generator(device=DEV).manual_seed(0)
cases = [("encoder stage 0: 96 -> 96", (81, 512, 896, 96)),
         ("encoder stage 1: 192 -> 192", (81, 256, 448, 192)),
         ("encoder stage 2: 384 -> 384", (41, 128, 224, 384))]
for name, xs in cases:
    T, H, W, C = xs
    x = torch.randn(T, H, W, C, device=DEV, generator=g).to(torch.bfloat16)
    wp = ops.prep_conv_weight(torch.randn(C, C, 3, 3, device=DEV, generator=g) / (9 * C) ** 0.5, torch.randn(C, device=DEV, generator=g))
    y = torch.empty(T, H // 2, W // 2, C, device=DEV, dtype=torch.bfloat16)
    alg = (x.numel() + y.numel()) * 2
    flop = 2.0 * 9 * C * C * y.numel() / C
    rec = {"conv": name, "in": list(xs), "algorithmic_GB": round(alg / 1e9, 2), "TFLOP": round(flop / 1e12, 3)}
    outs = {}
    for rnd in range(2):
        for mode, nm in ((0, "gather"), (1, "s2_nf1"), (2, "s2_nf2")):
            L.set_option("conv_s2", mode)
            rec.setdefault(nm + "_ms", []).append(round(timeit(lambda: ops.conv3d_cl(x, wp, (T, H // 2, W // 2), stride=(1, 2, 2), pad=(0, 0, 0), out=y)), 3))
            outs[nm] = y.clone()
    L.set_option("conv_s2", 1)
    for nm in ("gather", "s2_nf1", "s2_nf2"):
        rec[nm + "_TFLOPs"] = round(flop / min(rec[nm + "_ms"]) / 1e9, 0)
        rec[nm + "_GBps"] = round(alg / min(rec[nm + "_ms"]) / 1e6, 0)
    rec["max_abs_diff_nf1"] = float((outs["gather"].float() - outs["s2_nf1"].float()).abs().max())
    rec["nf1_equals_nf2"] = bool(torch.equal(outs["s2_nf1"], outs["s2_nf2"]))
    print(json.dumps(rec), flush=True)
    del x, y, outs

#!/usr/bin/env python
"""Same-process A/B of the decoder's three upsample convolutions (Resample: nearest 2x + 3 x 3, wan_vae.py:76-85) at BASELINE config 4's sizes:
the generated kt = 1 kernel (scail_conv4u_e0, option conv4 = 1) against the hipcc halo kernel (conv4 = 0).  One JSON line per shape."""
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
for (T, H, W, cin) in ((41, 64, 112, 384), (81, 128, 224, 384), (81, 256, 448, 192)):
    cout = cin // 2
    x = torch.randn(T, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (9 * cin) ** 0.5
    b = torch.randn(cout, device=DEV, generator=g)
    wp = ops.prep_conv_weight(w, b)
    y = torch.empty(T, 2 * H, 2 * W, cout, device=DEV, dtype=torch.bfloat16)
    fl = 2.0 * T * 4 * H * W * cout * 9 * cin
    rec = {"in": [T, H, W, cin], "out_channels": cout, "TFLOP": round(fl / 1e12, 2)}
    outs = {}
    for rnd in range(2):
        for mode, name in ((0, "halo"), (1, "conv4u")):
            L.set_option("conv4", mode)
            ms = timeit(lambda: ops.conv3d_cl(x, wp, (T, 2 * H, 2 * W), pad=(0, 1, 1), ups=True, out=y))
            rec.setdefault(name + "_ms", []).append(round(ms, 3))
            outs[name] = y.clone()
    L.set_option("conv4", 1)
    for name in ("halo", "conv4u"):
        rec[name + "_TFLOPs"] = round(fl / min(rec[name + "_ms"]) / 1e9, 0)
    rec["max_abs_diff"] = float((outs["halo"].float() - outs["conv4u"].float()).abs().max())
    print(json.dumps(rec), flush=True)
    del x, y, outs

# the decoder head CausalConv3d(96, 3, 3) at full resolution: the narrow-output kernel (scail_conv4n_e0) against the halo kernel (32-wide N tile)
T, H, W, cin, cout = 81, 512, 896, 96, 3
x = torch.randn(T, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
wp = ops.prep_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=DEV, generator=g) / (27 * cin) ** 0.5, torch.randn(cout, device=DEV, generator=g))
y = torch.empty(T, H, W, wp["N"], device=DEV, dtype=torch.bfloat16)
rec = {"head": [T, H, W, cin, cout], "algorithmic_GB": round((x.numel() + y.numel()) * 2 / 1e9, 2)}
outs = {}
for rnd in range(2):
    for (c4, cont), name in (((0, 1), "halo"), ((1, 0), "conv4n"), ((1, 1), "conv4n_continuation")):
        L.set_option("conv4", c4); L.set_option("conv4_cont", cont)
        ms = timeit(lambda: ops.conv3d_cl(x, wp, (T, H, W), out=y))
        rec.setdefault(name + "_ms", []).append(round(ms, 3))
        outs[name] = y.clone()
L.set_option("conv4", 1); L.set_option("conv4_cont", 1)
rec["continuation_identical"] = bool(torch.equal(outs["conv4n"], outs["conv4n_continuation"]))
rec["max_abs_diff"] = float((outs["halo"].float() - outs["conv4n"].float()).abs().max())
print(json.dumps(rec), flush=True)

# conv -> RMS_norm -> SiLU at the full-resolution 96-channel stage: scail_conv4f_e4 (one launch) against scail_conv4_e0 + rms_silu (two)
T, H, W, C = 21, 512, 896, 96
x = torch.randn(T, H, W, C, device=DEV, generator=g).to(torch.bfloat16)
wp = ops.prep_conv_weight(torch.randn(C, C, 3, 3, 3, device=DEV, generator=g) / (27 * C) ** 0.5, torch.randn(C, device=DEV, generator=g))
gam = 1 + 0.1 * torch.randn(C, device=DEV, generator=g)
y = torch.empty(T, H, W, C, device=DEV, dtype=torch.bfloat16)
rec = {"conv_norm": [T, H, W, C]}
for rnd in range(2):
    rec.setdefault("fused_e4_ms", []).append(round(timeit(lambda: ops.conv3d_cl_norm(x, wp, gam, out=y)), 3))
    rec.setdefault("conv_e0_ms", []).append(round(timeit(lambda: ops.conv3d_cl(x, wp, (T, H, W), out=y)), 3))
    rec.setdefault("conv_e0_plus_rms_silu_ms", []).append(round(timeit(lambda: ops.rms_silu(ops.conv3d_cl(x, wp, (T, H, W), out=y), gam, out=y)), 3))
print(json.dumps(rec), flush=True)

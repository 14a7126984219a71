#!/usr/bin/env python
"""Same-process A/B of the tile-continuation variants (option conv4_cont: scail_conv4c_e0 / e3 / e4) against the shipped generated kernels on the
shapes of the VAE: 96 output channels (one n tile: runs of frame pairs continue the rings), 192 / 384 (strided walk: only the useless
prefetch of a tile's last slice is dropped); the outputs must be bit-identical (same arithmetic, only the loads change)."""
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
for (T, H, W, cin, cout) in ((21, 512, 896, 96, 96), (81, 512, 896, 96, 96), (21, 256, 448, 192, 96), (81, 256, 448, 192, 192), (41, 128, 224, 384, 384),
                             (7, 40, 56, 96, 96)):
    x = torch.randn(T, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    wp = ops.prep_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=DEV, generator=g) / (27 * cin) ** 0.5, torch.randn(cout, device=DEV, generator=g))
    r = torch.randn(T, H, W, cout, device=DEV, generator=g).to(torch.bfloat16)
    gam = 1 + 0.1 * torch.randn(cout, device=DEV, generator=g)
    y = torch.empty(T, H, W, cout, device=DEV, dtype=torch.bfloat16)
    fl = 2.0 * T * H * W * cout * 27 * cin
    rec = {"shape": [T, H, W, cin, cout], "TFLOP": round(fl / 1e12, 2)}
    fns = {"e0": lambda: ops.conv3d_cl(x, wp, (T, H, W), out=y), "e3": lambda: ops.conv3d_cl(x, wp, (T, H, W), out=y, resid=r)}
    if cout == 96:
        fns["e4"] = lambda: ops.conv3d_cl_norm(x, wp, gam, out=y)
    for name, fn in fns.items():
        outs = []
        for rnd in range(2):
            for mode in (0, 1):
                L.set_option("conv4_cont", mode)
                rec.setdefault(f"{name}_{'cont' if mode else 'base'}_ms", []).append(round(timeit(fn), 3))
                if rnd == 0:
                    outs.append(y.clone())
        rec[f"{name}_identical"] = bool(torch.equal(outs[0], outs[1]))
    L.set_option("conv4_cont", 0)
    print(json.dumps(rec), flush=True)
    del x, r, y

#!/usr/bin/env python
"""A/B of the 3x3x3 convolution kernels of the Wan2.1 VAE on its three dominant shapes (channels-last, causal 'same'):
  python tools/conv_probe.py [--knobs 3,4,0] [--frames 21]
knob = conv_halo of scail_tune_set (0 gather kernel, 3 halo tile 8x16 x 1 frame, 4 halo tile 8x16 x 2 frames, ...)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import lib, ops as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--knobs", default="3,4")
ap.add_argument("--frames", type=int, default=21)
a = ap.parse_args()
lib.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for C, H, W in ((96, 512, 896), (192, 256, 448), (384, 128, 224)):
    T = a.frames
    x = torch.randn(T, H, W, C, device=dev, generator=g).to(torch.bfloat16)
    wp = O.prep_conv_weight(torch.randn(C, C, 3, 3, 3, device=dev, generator=g) * 0.02, torch.randn(C, device=dev, generator=g))
    out = torch.empty(T, H, W, C, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * H * W * C * C * 27
    ref = None
    for rnd in range(2):
        for v in [int(k) for k in a.knobs.split(",")]:
            lib.tune_set("conv_halo", v)
            f = lambda: O.conv3d_cl(x, wp, (T, H, W), out=out)
            f(); f(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            if ref is None:
                ref = out.clone()
            print(json.dumps(dict(C=C, H=H, W=W, T=T, knob=v, ms=sorted(ts)[2], tflops=fl / sorted(ts)[2] / 1e9,
                                  maxdiff_vs_first=float((out.float() - ref.float()).abs().max()))), flush=True)
# the 1x3x3 convolutions behind the nearest 2x upsample (decoder Resample stages)
for Cin, H, W in ((384, 128, 224), (192, 256, 448)):
    T, N = a.frames, Cin // 2
    x = torch.randn(T, H, W, Cin, device=dev, generator=g).to(torch.bfloat16)
    wp = O.prep_conv_weight(torch.randn(N, Cin, 3, 3, device=dev, generator=g) * 0.02, torch.randn(N, device=dev, generator=g))
    out = torch.empty(T, 2 * H, 2 * W, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * 4 * H * W * N * Cin * 9
    ref = None
    for rnd in range(2):
        for v in [int(k) for k in a.knobs.split(",")]:
            lib.tune_set("conv_halo", v)
            f = lambda: O.conv3d_cl(x, wp, (T, 2 * H, 2 * W), pad=(0, 1, 1), ups=True, out=out)
            f(); f(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            if ref is None:
                ref = out.clone()
            print(json.dumps(dict(ups=True, Cin=Cin, N=N, H_out=2 * H, W_out=2 * W, T=T, knob=v, ms=sorted(ts)[2], tflops=fl / sorted(ts)[2] / 1e9,
                                  maxdiff_vs_first=float((out.float() - ref.float()).abs().max()))), flush=True)
lib.tune_set("conv_halo", 4)

#!/usr/bin/env python
"""The dominant convolution launches of the Wan2.1 VAE on the product library (3x3x3 causal 'same', channels-last, 21 frames):
C = 96 @ 512x896, C = 192 @ 256x448, C = 384 @ 128x224, `iters` launches each: the process rocprofv3 --pmc wraps for the fabric traffic
of conv_halo_kernel (FETCH_SIZE / WRITE_SIZE in separate passes).  usage: python tools/conv_pmc_probe.py <C: 96 | 192 | 384> [iters] [conv4]   (one shape per run: the three share a kernel name;
"conv4": the generated kernel through the measurement build's knob, SCAIL_ABLATIONS=1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib, ops as O  # noqa: E402

which = int(sys.argv[1]) if len(sys.argv) > 1 else 96
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib.load()
if len(sys.argv) > 3 and sys.argv[3].startswith("conv4"):
    lib.tune_set("conv_halo", 11)
    if ":" in sys.argv[3]:
        lib.tune_set("conv4_kernel:" + sys.argv[3].split(":")[1], 0)
g = torch.Generator(device="cuda").manual_seed(0)
T = 21
for C, H, W in ((96, 512, 896), (192, 256, 448), (384, 128, 224)):
    if C != which:
        continue
    x = torch.randn(T, H, W, C, device="cuda", generator=g).to(torch.bfloat16)
    wp = O.prep_conv_weight(torch.randn(C, C, 3, 3, 3, device="cuda", generator=g) * 0.02, torch.randn(C, device="cuda", generator=g))
    out = torch.empty(T, H, W, C, device="cuda", dtype=torch.bfloat16)
    for _ in range(iters):
        O.conv3d_cl(x, wp, (T, H, W), out=out)
    torch.cuda.synchronize()
    del x, out

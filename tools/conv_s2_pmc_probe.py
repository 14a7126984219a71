#!/usr/bin/env python
"""(round 6) Fixed workload for rocprofv3 --pmc passes over Resample's stride-2 convolutions (96- and 192-channel encoder shapes of config 4, three launches
each): python tools/conv_s2_pmc_probe.py <conv_s2 option value: 0 gather kernel | 1 conv_s2_kernel>."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib as L, ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for (T, H, W, C) in [(81, 512, 896, 96), (81, 256, 448, 192)]:
    x = torch.randn(T, H, W, C, device=DEV, generator=g).to(torch.bfloat16)
    wp = ops.prep_conv_weight(torch.randn(C, C, 3, 3, device=DEV, generator=g) / (9 * C) ** 0.5, torch.randn(C, device=DEV, generator=g))
    y = torch.empty(T, H // 2, W // 2, C, device=DEV, dtype=torch.bfloat16)
    L.set_option("conv_s2", mode)
    for _ in range(3):
        ops.conv3d_cl(x, wp, (T, H // 2, W // 2), stride=(1, 2, 2), pad=(0, 0, 0), out=y)
    torch.cuda.synchronize()
    print(f"C={C}: algorithmic read {x.numel() * 2 / 1e9:.3f} GB, write {y.numel() * 2 / 1e9:.3f} GB", flush=True)

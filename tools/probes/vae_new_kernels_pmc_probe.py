#!/usr/bin/env python
"""Fixed workload for rocprofv3 --pmc passes over the round-4 VAE kernels at BASELINE config 4's sizes: two launches each of
scail_conv4u_e0 (81x256x448x192 -> 96 behind the 2x upsample), scail_conv4n_e0 (the RGB head 81x512x896x96 -> 3), scail_conv4f_e4
(21x512x896x96, conv + RMS_norm + SiLU), conv_direct_kernel<14, 3> (the stem) and <6, 6> (shortcut 96 -> 192).  Prints the algorithmic
bytes per launch (input + output + weights once) as JSON for the comparison with FETCH_SIZE / WRITE_SIZE."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
alg = {}


def run(name, x, wp, out_shape, cout_pad, **kw):
    y = torch.empty(*out_shape, cout_pad, device=DEV, dtype=torch.bfloat16)
    for _ in range(2):
        ops.conv3d_cl(x, wp, out_shape, out=y, **kw)
    alg[name] = (x.numel() + y.numel() + wp["w"].numel()) * 2
    torch.cuda.synchronize()


x = rn(81, 256, 448, 192).to(torch.bfloat16)
run("scail_conv4u_e0", x, ops.prep_conv_weight(rn(96, 192, 3, 3) / 41.6, rn(96)), (81, 512, 896), 96, pad=(0, 1, 1), ups=True)
del x
x = rn(81, 512, 896, 96).to(torch.bfloat16)
run("scail_conv4n_e0", x, ops.prep_conv_weight(rn(3, 96, 3, 3, 3) / 50.9, rn(3)), (81, 512, 896), 8)
del x
x = rn(21, 512, 896, 96).to(torch.bfloat16)
wp = ops.prep_conv_weight(rn(96, 96, 3, 3, 3) / 50.9, rn(96))
y = torch.empty(21, 512, 896, 96, device=DEV, dtype=torch.bfloat16)
for _ in range(2):
    ops.conv3d_cl_norm(x, wp, torch.ones(96, device=DEV), out=y)
alg["scail_conv4f_e4"] = (x.numel() + y.numel() + wp["w"].numel()) * 2
del x, y
x = torch.zeros(81, 512, 896, 8, device=DEV, dtype=torch.bfloat16)
x[..., :3] = rn(81, 512, 896, 3).to(torch.bfloat16)
run("conv_direct_kernel<14, 3>", x, ops.prep_conv_weight(rn(96, 3, 3, 3, 3) / 9.0, rn(96), cin_pad=8), (81, 512, 896), 96)
del x
x = rn(81, 256, 448, 96).to(torch.bfloat16)
run("conv_direct_kernel<6, 6>", x, ops.prep_conv_weight(rn(192, 96, 1, 1, 1) / 9.8, rn(192)), (81, 256, 448), 192)
torch.cuda.synchronize()
print(json.dumps({"algorithmic_bytes_per_launch": alg}))

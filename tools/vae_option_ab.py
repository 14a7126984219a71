#!/usr/bin/env python
"""Same-process A/B of a VAE kernel option (round 6: "conv4_resnorm" -- the residual + next-norm epilogues scail_conv4c_e5 / e6 / scail_conv4u_e7; "conv_s2" --
the stride-2 halo kernel of Resample's downsampling convolution): config 4's VAE encode / decode (81 x 512 x 896, random-init weights) with the option
off / on / off / on, min of `iters` runs each.  One JSON line per (direction, setting).
usage: python tools/vae_option_ab.py [iters] [option] [on-value]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib as L  # noqa: E402
from scail_amd.wan_vae import WanVAE_  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
OPT = sys.argv[2] if len(sys.argv) > 2 else "conv4_resnorm"
ON = int(sys.argv[3]) if len(sys.argv) > 3 else 1
L.load()
dev = "cuda"
m = WanVAE_(dim=96, z_dim=16, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
video = torch.rand(1, 3, 81, 512, 896, device=dev, generator=g) * 2 - 1
z = torch.randn(1, 16, 21, 64, 112, device=dev, generator=g)
outs = {}
for setting in (0, ON, 0, ON):
    L.set_option(OPT, setting)
    for name, fn, arg in (("encode", m.encode, video), ("decode", m.decode, z)):
        fn(arg)
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = fn(arg); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        rec = {"direction": name, OPT: setting, "ms_min": min(ts), "ms_all": [round(t, 2) for t in ts], "finite": bool(torch.isfinite(out).all())}
        if (name, ON - setting) in outs:
            o = outs[(name, ON - setting)]
            d = (out.float() - o.float()).abs()
            rec["max_abs_diff_vs_other_setting"] = float(d.max())
            rec["cosine_vs_other_setting"] = float(torch.nn.functional.cosine_similarity(out.flatten().float(), o.flatten().float(), dim=0))
        outs[(name, setting)] = out
        print(json.dumps(rec), flush=True)
L.set_option(OPT, ON)

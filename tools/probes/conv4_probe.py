#!/usr/bin/env python
"""GPU check + A/B of the generated 3x3x3 convolution kernels (scail_amd/asmgen/conv4.py) against the hipcc halo kernel.
Measurement build:  SCAIL_ABLATIONS=1 python tools/conv4_probe.py [--variants ,abl_dma,...] [--frames 21]"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import lib, ops as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="")
ap.add_argument("--frames", type=int, default=21)
ap.add_argument("--skip-check", action="store_true")
ap.add_argument("--prof", action="store_true", help="phase timers of the _prof variant (s_memtime), per tile")
ap.add_argument("--fit", action="store_true", help="per-tile / per-tap cost of conv4: N = 96, Cin = 32 .. 192 on the 512 x 896 shape")
a = ap.parse_args()
lib.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
variants = a.variants.split(",")


def setk(v):
    lib.tune_set("conv4_kernel" + (":" + v if v else ""), 0)


if not a.skip_check:
    # ragged tiles, odd frame count, residual, two N tiles; against torch fp32 on the bf16-rounded operands
    for (T, H, W, Cin, N, res) in ((3, 20, 18, 64, 96, False), (5, 33, 40, 32, 192, True), (2, 16, 16, 96, 96, True)):
        x = torch.randn(T, H, W, Cin, device=dev, generator=g).to(torch.bfloat16)
        w = torch.randn(N, Cin, 3, 3, 3, device=dev, generator=g) * 0.05
        b = torch.randn(N, device=dev, generator=g)
        wp = O.prep_conv_weight(w, b)
        r = torch.randn(T, H, W, N, device=dev, generator=g).to(torch.bfloat16) if res else None
        xr = F.pad(x.float().permute(3, 0, 1, 2)[None], (1, 1, 1, 1, 2, 0))
        ref = F.conv3d(xr, w.to(torch.bfloat16).float(), b)[0].permute(1, 2, 3, 0)
        if res:
            ref = ref + r.float()
        for mode in (10, 11):
            lib.tune_set("conv_halo", mode)
            for var in (variants if mode == 11 else [""]):
                if "abl" in var or (res and var):
                    continue
                setk(var)
                out = torch.full((T, H, W, N), float("nan"), device=dev, dtype=torch.bfloat16)
                O.conv3d_cl(x, wp, (T, H, W), out=out, **(dict(resid=r) if res else {}))
                torch.cuda.synchronize()
                err = (out.float() - ref).abs()
                print(json.dumps(dict(check=[T, H, W, Cin, N, res], conv4=mode == 11, variant=var, max_err=float(err.max()), nan=int(torch.isnan(out.float()).sum()),
                                      ok=bool(err.max() < 0.06))), flush=True)
    setk("")

if a.prof:
    lib.tune_set("conv_halo", 11)
    names = ["setup", "dma_issue", "acc_to_regs", "wait_first_loads", "stores", "acc_init_barrier", "slices", "closing_barrier"]
    for C, H, W in ((96, 512, 896), (192, 256, 448), (384, 128, 224)):
        T = a.frames
        x = torch.randn(T, H, W, C, device=dev, generator=g).to(torch.bfloat16)
        wp = O.prep_conv_weight(torch.randn(C, C, 3, 3, 3, device=dev, generator=g) * 0.02, torch.randn(C, device=dev, generator=g))
        out = torch.empty(T, H, W, C, device=dev, dtype=torch.bfloat16)
        dbg = torch.zeros_like(out)
        tiles = (T + 1) // 2 * (H // 16) * (W // 16) * (C // 96)
        for pv in ("prof",):
            setk(pv)
            O.conv3d_cl(x, wp, (T, H, W), out=out, resid=dbg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dbg.zero_()
            e0.record(); O.conv3d_cl(x, wp, (T, H, W), out=out, resid=dbg); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            d = dbg.view(-1).view(torch.int32)[:256 * 8].view(256, 8).double()
            per_wg_tiles = tiles / 256.0
            mean = d.mean(0) / per_wg_tiles
            total = float(d.sum(1).mean())
            print(json.dumps(dict(prof=pv, C=C, ms=ms, tiles_per_wg=per_wg_tiles, total_cycles_per_wg=total, mhz=total / ms / 1e3,
                                  cycles_per_tile={n: round(float(v)) for n, v in zip(names, mean)},
                                  taps_per_tile=27 * C // 32, cycles_per_tap=float(mean[6]) / (27 * C // 32))), flush=True)
        setk("")
        del x, out, dbg
    lib.tune_set("conv_halo", 10)
    sys.exit(0)

if a.fit:
    lib.tune_set("conv_halo", 11)
    T, H, W, N = a.frames, 512, 896, 96
    tiles = (T + 1) // 2 * (H // 16) * (W // 16)
    pts = []
    for Cin in (32, 64, 96, 128, 192):
        x = torch.randn(T, H, W, Cin, device=dev, generator=g).to(torch.bfloat16)
        wp = O.prep_conv_weight(torch.randn(N, Cin, 3, 3, 3, device=dev, generator=g) * 0.02, torch.randn(N, device=dev, generator=g))
        out = torch.empty(T, H, W, N, device=dev, dtype=torch.bfloat16)
        f = lambda: O.conv3d_cl(x, wp, (T, H, W), out=out)
        f(); f(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[2]
        per_tile_us = ms * 1e3 / (tiles / 256.0)
        pts.append((Cin // 32, per_tile_us))
        print(json.dumps(dict(fit=True, Cin=Cin, ms=ms, tiles=tiles, us_per_tile_per_cu=per_tile_us, tflops=2.0 * T * H * W * N * Cin * 27 / ms / 1e9)), flush=True)
        del x, out
    n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts); sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
    print(json.dumps(dict(fit=True, us_per_slice=slope, us_per_tap=slope / 27, us_fixed_per_tile=icpt,
                          note="48 MFMAs of 16 cycles per tap = 768 cycles = 0.32 us at 2.4 GHz / 0.40 us at 1.9 GHz")), flush=True)
    lib.tune_set("conv_halo", 10)
    sys.exit(0)

for C, H, W in ((96, 512, 896), (192, 256, 448), (384, 128, 224)):
    T = a.frames
    x = torch.randn(T, H, W, C, device=dev, generator=g).to(torch.bfloat16)
    wp = O.prep_conv_weight(torch.randn(C, C, 3, 3, 3, device=dev, generator=g) * 0.02, torch.randn(C, device=dev, generator=g))
    out = torch.empty(T, H, W, C, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * H * W * C * C * 27
    ref = None
    for mode, var in [(10, "")] + [(11, v) for v in variants]:
        lib.tune_set("conv_halo", mode)
        setk(var)
        f = lambda: O.conv3d_cl(x, wp, (T, H, W), out=out)
        out.zero_()
        f(); f(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        if ref is None:
            ref = out.clone()
        print(json.dumps(dict(C=C, H=H, W=W, T=T, kernel="halo (hipcc)" if mode == 10 else "conv4" + var, ms=sorted(ts)[2], tflops=fl / sorted(ts)[2] / 1e9,
                              maxdiff_vs_halo=float((out.float() - ref.float()).abs().max()))), flush=True)
    # with the residual epilogue (the second convolution of a ResidualBlock)
    r = torch.randn(T, H, W, C, device=dev, generator=g).to(torch.bfloat16)
    for mode in (10, 11):
        lib.tune_set("conv_halo", mode)
        setk("")
        f = lambda: O.conv3d_cl(x, wp, (T, H, W), out=out, resid=r)
        f(); f(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print(json.dumps(dict(C=C, H=H, W=W, T=T, resid=True, kernel="halo (hipcc)" if mode == 10 else "conv4", ms=sorted(ts)[2], tflops=fl / sorted(ts)[2] / 1e9)), flush=True)
lib.tune_set("conv_halo", 10)
setk("")

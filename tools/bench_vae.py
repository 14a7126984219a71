#!/usr/bin/env python
"""BASELINE.json configs[3]: Wan2.1 3D causal VAE encode + decode only, 512p x 81 f, one MI355X.
Prints one JSON line per direction: time, algorithmic TFLOP/s (MFMA roofline) and algorithmic GB/s (HBM
roofline; every conv reads its input once and writes its output once, norm/SiLU fused -- SURVEY.md 8d:
encode 188.3 TFLOP / 148.7 GB, decode 316.5 TFLOP / 229.4 GB).  Random-init weights, synthetic video."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib  # noqa: E402
from scail_amd.wan_vae import WanVAE_  # noqa: E402

ALG = {"encode": (188.3e12, 148.7e9), "decode": (316.5e12, 229.4e9)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=81)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=896)
    ap.add_argument("--iters", type=int, default=2)
    a = ap.parse_args()
    lib.load()
    dev = "cuda"
    m = WanVAE_(dim=96, z_dim=16, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    video = torch.rand(1, 3, a.frames, a.height, a.width, device=dev, generator=g) * 2 - 1
    z = torch.randn(1, 16, 1 + (a.frames - 1) // 4, a.height // 8, a.width // 8, device=dev, generator=g)
    full = (a.frames, a.height, a.width) == (81, 512, 896)
    vox = a.frames * a.height * a.width
    for name, fn, arg in (("encode", m.encode, video), ("decode", m.decode, z)):
        fn(arg)                                  # warm-up (weight prep, allocator)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = fn(arg); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        scale = vox / (81 * 512 * 896)
        fl, by = ALG[name][0] * scale, ALG[name][1] * scale
        print(json.dumps({"workload": f"Wan2.1 VAE {name} {a.frames}x{a.height}x{a.width}", "ms": ms, "pixels_per_s": vox / ms * 1e3,
                          "alg_tflops": fl / ms / 1e9, "mfma_frac": fl / ms / 1e9 / 2500, "alg_GBps": by / ms / 1e6,
                          "hbm_frac": by / ms / 1e6 / 8000, "finite": bool(torch.isfinite(out).all()),
                          "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9, "exact_config4": full}), flush=True)


if __name__ == "__main__":
    main()

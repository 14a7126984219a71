#!/usr/bin/env python
"""Kernel microbenchmarks on random data (HIP events on the launch stream), one JSON line per case.
  python tools/microbench.py attn [--L 48832] [--heads 8] [--iters 5]
  python tools/microbench.py gemm [--M 97664 --N 5120 --K 5120] [--epi 0]
Used under rocprofv3 --pmc for counter collection (profiles/).
Timing-ablation variants (gemm_tile >= 1000, attn_variant 18 / 34 / 50 / swp sub-code 5; wrong results on purpose) need the
measurement build: SCAIL_ABLATIONS=1 python -m scail_amd.build, then run this tool with SCAIL_ABLATIONS=1."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import lib, ops  # noqa: E402


def timeit(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["attn", "gemm", "rows"])
    ap.add_argument("--vendor", action="store_true", help="also time torch's library GEMM (yardstick)")
    ap.add_argument("--L", type=int, default=48832)
    ap.add_argument("--Lk", type=int, default=None)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--M", type=int, default=97664)
    ap.add_argument("--N", type=int, default=5120)
    ap.add_argument("--K", type=int, default=5120)
    ap.add_argument("--epi", type=int, default=0)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--variants", type=str, default="0")
    a = ap.parse_args()
    lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
    if a.what == "attn":
        D = a.heads * 128
        Lk = a.Lk or a.L
        qkv = rn(a.B, a.L, 3 * D)
        q = qkv[..., :D]
        if Lk == a.L:
            k, v = qkv[..., D:2 * D], qkv[..., 2 * D:]
        else:
            k, v = rn(a.B, Lk, D), rn(a.B, Lk, D)
        vt = ops.transpose_v(v, a.heads)
        out = torch.empty(a.B, a.L, D, device=dev, dtype=torch.bfloat16)
        fl = 4.0 * a.L * Lk * 128 * a.heads * a.B
        ref = None
        for rnd_ in range(2):                       # interleaved rounds: within-probe A/B
            for var in [int(x) for x in a.variants.split(",")]:
                lib.tune_set("attn_variant", var)
                med, best = timeit(lambda: ops.flash_attn(q, k, vt, out=out), a.iters)
                if ref is None:
                    ref = out.clone()
                diff = float((out.float() - ref.float()).abs().max())
                rec = dict(case="attn", variant=var, B=a.B, heads=a.heads, Lq=a.L, Lk=Lk, ms=med, ms_min=best,
                           tflops=fl / med / 1e9, maxdiff_vs_first=diff)
                if var & (1 << 20):         # clock-stamped: one more launch, workgroup lifetimes in shader cycles
                    import ctypes
                    buf = (ctypes.c_ulonglong * 2)()
                    lib.call("scail_debug_cycles", None, 1)
                    ops.flash_attn(q, k, vt, out=out)
                    torch.cuda.synchronize()
                    lib.call("scail_debug_cycles", ctypes.cast(buf, ctypes.c_void_p), 0)
                    ntile = (Lk + 63) // 64
                    rec.update(wg=int(buf[1]), cycles_per_wg=buf[0] / max(buf[1], 1), cycles_per_tile=buf[0] / max(buf[1], 1) / ntile,
                               ghz=buf[0] / 256.0 / (med * 1e6))
                print(json.dumps(rec))
    elif a.what == "gemm":
        x, w = rn(a.M, a.K), rn(a.N, a.K) * 0.02
        b = torch.randn(a.N, device=dev)
        y = torch.empty(a.M, a.N, device=dev, dtype=torch.bfloat16)
        kw = {}
        if a.epi == 3:
            kw = dict(resid=y, gate=torch.randn(2, a.N, device=dev), rows_per_batch=a.M // 2)
        fl = 2.0 * a.M * a.N * a.K
        if a.vendor:      # yardstick only (never on the product path): the vendor library through torch
            bb = b.to(torch.bfloat16)
            med, best = timeit(lambda: torch.nn.functional.linear(x, w, bb), a.iters)
            print(json.dumps(dict(case="gemm", tile="vendor(torch F.linear)", M=a.M, N=a.N, K=a.K, ms=med, ms_min=best, tflops=fl / med / 1e9)))
        for rnd_ in range(2):
            for var in [int(z) for z in a.variants.split(",")]:
                lib.tune_set("gemm_tile", var)
                med, best = timeit(lambda: ops.gemm(x, w, b, out=y, epilogue=a.epi, **kw), a.iters)
                rec = dict(case="gemm", tile=var, M=a.M, N=a.N, K=a.K, epi=a.epi, ms=med, ms_min=best, tflops=fl / med / 1e9)
                if 1300 <= var < 1400:      # clock-stamped variants: one more launch, span in shader cycles vs wall time
                    import ctypes
                    buf = (ctypes.c_ulonglong * 2)()
                    lib.call("scail_debug_cycles", None, 1)
                    ops.gemm(x, w, b, out=y, epilogue=a.epi, **kw)
                    torch.cuda.synchronize()
                    lib.call("scail_debug_cycles", ctypes.cast(buf, ctypes.c_void_p), 0)
                    cyc = buf[0] / 256.0            # busy cycles per CU (1 workgroup per CU at a time)
                    rec.update(wg=int(buf[1]), cycles_per_wg=buf[0] / max(buf[1], 1), ghz=cyc / (med * 1e6))
                print(json.dumps(rec))
    else:
        D = 5120
        x = rn(2, a.L, D)
        sh, sc = torch.randn(2, D, device=dev), torch.randn(2, D, device=dev)
        y = torch.empty_like(x)
        med, best = timeit(lambda: ops.ln_modulate(x, sh, sc, out=y), a.iters)
        print(json.dumps(dict(case="ln_modulate", rows=2 * a.L, D=D, ms=med, GBps=2 * x.numel() * 2 / med / 1e6)))
        w = torch.randn(D, device=dev)
        med, best = timeit(lambda: ops.rmsnorm_rope(x, w), a.iters)
        print(json.dumps(dict(case="rmsnorm", rows=2 * a.L, D=D, ms=med, GBps=2 * x.numel() * 2 / med / 1e6)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""(round 5) the conditioning's K / V projections (text: 1024 x 10240 x 5120, CLIP: 257 x 10240 x 5120) and neighbours: generated 4-wave
kernel (option gemm4 = 1, eligible from 512 rows x >= 128 tiles on) against the hipcc kernels (gemm4 = 0), same process."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import lib, ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


g = torch.Generator(device=DEV).manual_seed(0)
for M, N, K in [(1024, 10240, 5120), (512, 10240, 5120), (768, 5120, 5120), (1024, 5120, 4096), (1536, 5120, 5120), (257, 10240, 5120), (2048, 10240, 5120)]:
    x = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device=DEV, generator=g)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    rec = {"M": M, "N": N, "K": K}
    outs = []
    for name, on in (("gemm4", 1), ("hipcc", 0)):
        lib.set_option("gemm4", on)
        rec[name + "_kernel_for"] = int(lib.load().scail_gemm_kernel_for(K, N, 0, M, N, K, lib.EPI_BIAS))
        ms = timeit(lambda: ops.gemm(x, w, b, out=y))
        outs.append(y.clone())
        rec[name + "_us"] = ms * 1e3
        rec[name + "_TFLOPs"] = 2.0 * M * N * K / ms / 1e9
    lib.set_option("gemm4", 1)
    rec["max_abs_diff"] = float((outs[0].float() - outs[1].float()).abs().max())
    print(json.dumps(rec), flush=True)

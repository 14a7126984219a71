#!/usr/bin/env python
"""Where does the residual epilogue's time go?  Product library, M = 97 664, N = 5 120: (1) the bias-only kernel at K = 5 120 / 10 240 /
20 480 (fixed per-tile cost from t(K)); (2) the residual kernels (e4: no gate, e3: gate) at K = 5 120 with the residual read in place,
from a separate buffer, and from ONE row broadcast to every row (stride 0: always L2-resident) -- the last one takes the HBM / fabric
latency of the residual reads out.  One JSON line per case."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import lib as L, ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=9):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
M, N = 97664, 5120
b = rn(N)
gate = rn(2, N)
for K in (5120, 10240, 20480):
    x = rn(M, K).to(torch.bfloat16)
    w = (rn(N, K) * 0.02).to(torch.bfloat16)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm(x, w, b, out=y))
    print(json.dumps({"case": "bias only (e0)", "K": K, "ms": ms, "TFLOPs": 2.0 * M * N * K / ms / 1e9}), flush=True)
    if K == 5120:
        r_full = rn(M, N).to(torch.bfloat16)
        r_row = rn(1, N).to(torch.bfloat16).expand(M, N)
        for name, kw in (("resid (e4)", {}), ("resid + gate (e3)", dict(gate=gate, rows_per_batch=M // 2))):
            for how, res in (("in place (the step's call)", None), ("separate buffer", r_full), ("one row broadcast (L2-resident)", r_row)):
                y.copy_(r_full)
                ms = timeit(lambda: ops.gemm(x, w, b, out=y, epilogue=L.EPI_RESID, resid=(y if res is None else res), **kw))
                print(json.dumps({"case": name, "residual": how, "K": K, "ms": ms, "TFLOPs": 2.0 * M * N * K / ms / 1e9}), flush=True)
    del x, w, y

#!/usr/bin/env python
"""Fixed cost of a generated-GEMM launch: time scail_gemm_bf16 (bias epilogue, K = 5120, N = 4096 = 16 n tiles) at M = 4096 r rows, i.e. exactly
r rounds of 256 tiles on 256 CUs, r = 1, 2, 3, 4, 6, 8, 16, 24: ms(r) = fixed + r * round.  The intercept is what a rank-sized GEMM of the
sequence-parallel path (4-11 rounds) pays per launch and a full-size one (30-80 rounds) amortises.  Also: the same launches back to back
WITHOUT events in between (N launches in one timed region), which separates the kernel's own ramp / drain from launch gaps."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import ops  # noqa: E402

DEV = "cuda"
K, N = 5120, 4096
g = torch.Generator(device=DEV).manual_seed(0)
w = (torch.randn(N, K, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
b = torch.zeros(N, device=DEV)


def timeit(fn, iters=10, reps=1):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, c in ev:
        a.record()
        for _ in range(reps):
            fn()
        c.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(c) / reps for a, c in ev)
    return ts[len(ts) // 2]


out = []
for r in (1, 2, 3, 4, 6, 8, 16, 24):
    M = 4096 * r
    x = (torch.randn(M, K, device=DEV, generator=g)).to(torch.bfloat16)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    one = timeit(lambda: ops.gemm(x, w, b, out=y))
    ten = timeit(lambda: ops.gemm(x, w, b, out=y), reps=10)
    out.append({"rounds": r, "M": M, "ms_single": round(one, 4), "ms_in_a_train_of_10": round(ten, 4), "TFLOPs_single": round(2.0 * M * N * K / one / 1e9, 0)})
    print(json.dumps(out[-1]), flush=True)
# least squares ms = a + b r
import numpy as np
rs = np.array([o["rounds"] for o in out], float)
for key in ("ms_single", "ms_in_a_train_of_10"):
    ys = np.array([o[key] for o in out])
    A = np.stack([np.ones_like(rs), rs], 1)
    (a, bb), *_ = np.linalg.lstsq(A, ys, rcond=None)
    print(json.dumps({"fit": key, "fixed_ms": round(float(a), 4), "ms_per_round": round(float(bb), 4)}))

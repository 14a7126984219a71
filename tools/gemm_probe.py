#!/usr/bin/env python
"""Product library: the four per-token GEMMs of a config-2 layer with the generated 4-wave kernels (default) and with the
hipcc kernels of csrc/gemm.hip (scail_set_option("gemm4", 0)), plus the vendor library (torch F.linear) as a yardstick.
One JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib as L, ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=7):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
M = 97664
for (N, K, epi, tag) in ((15360, 5120, L.EPI_BIAS, "qkv"), (5120, 5120, L.EPI_RESID, "out-proj + gate/resid"),
                         (13824, 5120, L.EPI_GELU_TANH, "mlp up + gelu"), (5120, 13824, L.EPI_RESID, "mlp down + gate/resid")):
    x = rn(M, K).to(torch.bfloat16)
    w = (rn(N, K) * 0.02).to(torch.bfloat16)
    b = rn(N)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    gate = rn(2, N)
    kw = dict(resid=y, gate=gate, rows_per_batch=M // 2) if epi == L.EPI_RESID else {}
    fl = 2.0 * M * N * K
    out = {"shape": [M, N, K], "what": tag}
    for on, name in ((1, "gemm4"), (0, "q8")):
        L.set_option("gemm4", on)
        ms = timeit(lambda: ops.gemm(x, w, b, out=y, epilogue=epi, **kw))
        out[name + "_ms"] = ms
        out[name + "_TFLOPs"] = fl / ms / 1e9
    L.set_option("gemm4", 1)
    ms = timeit(lambda: torch.nn.functional.linear(x, w))
    out["vendor_TFLOPs"] = fl / ms / 1e9
    print(json.dumps(out), flush=True)
    del x, w, y

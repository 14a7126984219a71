#!/usr/bin/env python
"""Measurement build: a variant set of the generated GEMM that exists for all four epilogues (argv[1]: "pst" = persistent workgroups, "part" = the
partial-line epilogue the kernels had before the LDS staging, "stgnt" = staged epilogue with non-temporal stores) against the shipped kernels, same process, on the six per-token GEMMs
of a config-2 block and on rank-sized ones; outputs compared bit for bit.  One JSON line per case."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scail_amd import lib as L, ops  # noqa: E402

DEV = "cuda"
VAR = sys.argv[1] if len(sys.argv) > 1 else "pst"


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
D, FF = 5120, 13824
for M in (97664, 6104, 2048 + 136):
    for (N, K, epi, tag) in ((3 * D, D, L.EPI_BIAS, "qkv"), (D, D, L.EPI_RESID, "out-proj + gate/resid"), (D, D, L.EPI_BIAS, "cross q"),
                             (D, D, L.EPI_RESID, "cross out + resid"), (FF, D, L.EPI_GELU_TANH, "mlp up + gelu"), (D, FF, L.EPI_RESID, "mlp down + gate/resid")):
        x = rn(M, K).to(torch.bfloat16)
        w = (rn(N, K) * 0.02).to(torch.bfloat16)
        b = rn(N)
        h0 = rn(M, N).to(torch.bfloat16)
        gate = rn(2, N)
        outs, res = {}, {"M": M, "shape": [M, N, K], "what": tag}
        for suffix in ("", VAR, "", VAR):
            L.tune_set("gemm4_kernel" + (":" + suffix if suffix else ""), 0)
            y = h0.clone()
            kw = {}
            if epi == L.EPI_RESID:
                kw = dict(resid=y, rows_per_batch=(M + 1) // 2 if "gate" in tag else 0, gate=gate if "gate" in tag else None)
            ops.gemm(x, w, b, out=y, epilogue=epi, **kw)
            torch.cuda.synchronize()
            outs[suffix or "shipped"] = y.clone()

            def call():
                ops.gemm(x, w, b, out=y, epilogue=epi, **kw)
            ms = timeit(call)
            key = (suffix or "shipped")
            res[key + "_TFLOPs"] = max(res.get(key + "_TFLOPs", 0.0), 2.0 * M * N * K / ms / 1e9)
        L.tune_set("gemm4_kernel", 0)
        res["variant"] = VAR
        res["bit_identical"] = bool(torch.equal(outs["shipped"], outs[VAR]))
        res["finite"] = bool(torch.isfinite(outs[VAR].float()).all())
        res["gain_pct"] = 100.0 * (res[VAR + "_TFLOPs"] / res["shipped_TFLOPs"] - 1.0)
        print(json.dumps(res), flush=True)
        del x, w, h0, y

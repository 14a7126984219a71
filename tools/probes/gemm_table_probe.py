#!/usr/bin/env python
"""Measurement build (SCAIL_ABLATIONS=1): the tile -> XCD assignment of the generated GEMM's order table (csrc/gemm.hip gemm4_table,
knob "gemm4_table") x the tile-group height (knob "gemm_group_m") on the four per-token GEMM shapes of a config-2 layer, interleaved
A/B in one process (HIP events, median of 7).  One JSON line per (shape, mode, group)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
shapes = ((15360, 5120, L.EPI_BIAS, "qkv"), (5120, 5120, L.EPI_RESID, "out-proj + gate/resid"),
          (13824, 5120, L.EPI_GELU_TANH, "mlp up + gelu"), (5120, 13824, L.EPI_RESID, "mlp down + gate/resid"))
combos = [(0, 4), (1, 4), (2, 4), (0, 2), (1, 2), (2, 2), (0, 6), (1, 6), (1, 8), (0, 4)]
for (N, K, epi, tag) in shapes:
    x = rn(M, K).to(torch.bfloat16)
    w = (rn(N, K) * 0.02).to(torch.bfloat16)
    b = rn(N)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    gate = rn(2, N)
    kw = dict(resid=y, gate=gate, rows_per_batch=M // 2) if epi == L.EPI_RESID else {}
    fl = 2.0 * M * N * K
    for mode, grp in combos:
        L.tune_set("gemm4_table", mode)
        L.tune_set("gemm_group_m", grp)
        ms = timeit(lambda: ops.gemm(x, w, b, out=y, epilogue=epi, **kw))
        print(json.dumps({"shape": [M, N, K], "what": tag, "table_mode": mode, "group_m": grp, "ms": ms, "TFLOPs": fl / ms / 1e9}), flush=True)
    L.tune_set("gemm4_table", 1)
    L.tune_set("gemm_group_m", 4)
    del x, w, y

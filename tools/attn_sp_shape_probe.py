#!/usr/bin/env python
"""Measurement build: the self-attention launch of ONE sequence-parallel rank (Ulysses, N ranks: heads / N heads of this rank, all
N * Ltok tokens, q / k in contiguous (L, Dn) matrices, one CFG element per launch) against the single-rank launch, per kernel variant.
usage: SCAIL_ABLATIONS=1 python tools/attn_sp_shape_probe.py [N ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib, ops  # noqa: E402

DEV = "cuda"
L = 48832


def timeit(fn, iters=5):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


lib.load()
g = torch.Generator(device=DEV).manual_seed(0)
for N in [int(a) for a in (sys.argv[1:] or ["8", "4", "1"])]:
    H = 40 // N
    B = 2 if N == 1 else 1
    D = H * 128
    q = (torch.randn(B, L, D, device=DEV, generator=g) * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
    k = torch.randn(B, L, D, device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn(B, L, D, device=DEV, generator=g).to(torch.bfloat16)
    vt = ops.transpose_v(v, H)
    out = torch.empty(B, L, D, device=DEV, dtype=torch.bfloat16)
    fl = 4.0 * L * L * 128 * H * B
    for var in ("", "m16f_noopt", "m16f_opt_db"):
        lib.tune_set("attn4_kernel" + (":" + var if var else ""), 0)
        for xcd in (1, 0):
            lib.tune_set("attn4_xcd", xcd)
            ms = timeit(lambda: ops.flash_attn(q, k, vt, out=out, q_prescaled=True))
            print(json.dumps({"ranks": N, "B": B, "heads": H, "workgroups": B * H * 191, "rounds_of_256": B * H * 191 / 256, "variant": var or "shipped",
                              "xcd_aware_ids": xcd, "ms": ms, "TFLOPs": fl / ms / 1e9}), flush=True)
    lib.tune_set("attn4_xcd", 1)
    lib.tune_set("attn4_kernel", 0)
    del q, k, v, vt, out

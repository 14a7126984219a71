#!/usr/bin/env python
"""The self-attention launch of ONE sequence-parallel rank (Ulysses, N ranks: heads / N heads of this rank, all N * Ltok tokens, q / k in
contiguous (L, Dn) matrices, one CFG element per launch; all-gather at 2 ranks: 40 heads x L / 2 queries) against the single-rank launch,
in ONE process, for every launch shape the product library can choose: query-tile height 256 / 192 rows (option "attn4_rows";
scail_attn4_m16f / scail_attn4_m16f_q3) x workgroup-id decode XCD-aware / plain (option "attn4_xcd").  Reports the per-tile cost of the
192-row tile relative to the 256-row tile (the constant of csrc/attn.hip attn4_pick_rows) and what the rounds model predicts.
usage: python tools/attn_sp_shape_probe.py [N ...]"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib, ops  # noqa: E402

DEV = "cuda"
L = 48832
CUS = torch.cuda.get_device_properties(0).multi_processor_count


def timeit(fn, iters=7):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


lib.load()
g = torch.Generator(device=DEV).manual_seed(0)
for N in [int(a) for a in (sys.argv[1:] or ["8", "4", "2", "1"])]:
    allgather = N == 2
    H = 40 if allgather else 40 // N
    B = 2 if N == 1 else 1
    Lq = L // N if allgather else L
    D = H * 128
    q = (torch.randn(B, Lq, D, device=DEV, generator=g) * ops.ATTN_LOG2_SCALE).to(torch.bfloat16)
    k = torch.randn(B, L, D, device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn(B, L, D, device=DEV, generator=g).to(torch.bfloat16)
    vt = ops.transpose_v(v, H)
    out = torch.empty(B, Lq, D, device=DEV, dtype=torch.bfloat16)
    fl = 4.0 * Lq * L * 128 * H * B
    res = {}
    for rows in (256, 192):
        for xcd in (1, 0):
            lib.set_option("attn4_rows", rows)
            lib.set_option("attn4_xcd", xcd)
            ms = timeit(lambda: ops.flash_attn(q, k, vt, out=out, q_prescaled=True))
            wgs = B * H * math.ceil(Lq / rows)
            res[(rows, xcd)] = ms
            print(json.dumps({"ranks": N, "B": B, "heads": H, "Lq": Lq, "rows": rows, "xcd_aware_ids": xcd, "workgroups": wgs, "rounds": wgs / CUS,
                              "ms": ms, "TFLOPs": fl / ms / 1e9, "ms_per_round": ms / math.ceil(wgs / CUS)}), flush=True)
    lib.set_option("attn4_rows", 0)
    lib.set_option("attn4_xcd", 1)
    auto = timeit(lambda: ops.flash_attn(q, k, vt, out=out, q_prescaled=True))
    r4, r3 = math.ceil(B * H * math.ceil(Lq / 256) / CUS), math.ceil(B * H * math.ceil(Lq / 192) / CUS)
    print(json.dumps({"ranks": N, "planned_shape": int(lib.load().scail_flash_attn_rows_for(B, H, Lq)), "planned_ms": auto, "best_single_ms": min(res.values()),
                      "gain_planned_vs_best_single": min(res.values()) / auto,
                      "tile_cost_192_over_256": (res[(192, 1)] / r3) / (res[(256, 1)] / r4), "rounds_256": r4, "rounds_192": r3,
                      "gain_192_vs_256": res[(256, 1)] / res[(192, 1)], "gain_xcd_256": res[(256, 0)] / res[(256, 1)], "gain_xcd_192": res[(192, 0)] / res[(192, 1)]}),
          flush=True)
    del q, k, v, vt, out

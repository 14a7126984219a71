#!/usr/bin/env python
"""The six per-token GEMMs of ONE config-2 transformer block on the product library (shapes, epilogues and operand layouts as
csrc/dit_step.hip issues them), `iters` times: the process rocprofv3 --pmc wraps to get the fabric traffic of scail_gemm4_e*
per launch (FETCH_SIZE / WRITE_SIZE in separate passes).  usage: python tools/gemm_layer_pmc_probe.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib as L, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L.load()
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
M, D, FF = 97664, 5120, 13824
xn = rn(M, D).to(torch.bfloat16)
h = rn(M, D).to(torch.bfloat16)
qkv = torch.empty(M, 3 * D, device="cuda", dtype=torch.bfloat16)
att = rn(M, D).to(torch.bfloat16)
ff = torch.empty(M, FF, device="cuda", dtype=torch.bfloat16)
W = {n: (rn(*s) * 0.02).to(torch.bfloat16) for n, s in (("qkv", (3 * D, D)), ("o", (D, D)), ("cq", (D, D)), ("co", (D, D)), ("w1", (FF, D)), ("w2", (D, FF)))}
b = {n: rn(w.shape[0]) for n, w in W.items()}
gate = rn(2, 6 * D)
for _ in range(iters):
    ops.gemm(xn, W["qkv"], b["qkv"], out=qkv)
    ops.gemm(att, W["o"], b["o"], out=h, epilogue=L.EPI_RESID, resid=h, gate=gate[:, 2 * D:3 * D], rows_per_batch=M // 2)
    ops.gemm(xn, W["cq"], b["cq"], out=qkv[:, :D])
    ops.gemm(att, W["co"], b["co"], out=h, epilogue=L.EPI_RESID, resid=h)
    ops.gemm(xn, W["w1"], b["w1"], out=ff, epilogue=L.EPI_GELU_TANH)
    ops.gemm(ff, W["w2"], b["w2"], out=h, epilogue=L.EPI_RESID, resid=h, gate=gate[:, 5 * D:], rows_per_batch=M // 2)
torch.cuda.synchronize()

#!/usr/bin/env python
"""One GEMM kernel family on the QKV shape of the step, a few launches: the process rocprofv3 --pmc wraps to compare the kernels of
csrc/gemm.hip (q8), gemm4 and gemm8.  usage: python tools/gemm_pmc_probe.py <0|4|8> [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import lib, ops  # noqa: E402

mode = int(sys.argv[1])
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib.load()
lib.tune_set("gemm4", mode)
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 97664, 15360, 5120
x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
b = torch.randn(N, device="cuda", generator=g)
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(iters):
    ops.gemm(x, w, b, out=y)
torch.cuda.synchronize()

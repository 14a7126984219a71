#!/usr/bin/env python
"""The vendor library (torch F.linear) on the QKV shape of config 2, a few launches: target of PMC passes (tools/run_vendor_pmc.sh)
that put the clock / MFMA-busy / traffic of the vendor's kernel beside the same counters of our q8 kernel."""
import sys

import torch

M, N, K = 97664, 15360, 5120
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    y = torch.nn.functional.linear(x, w)
torch.cuda.synchronize()

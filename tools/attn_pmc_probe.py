#!/usr/bin/env python
"""The self-attention launch of config 2 (B = 2, 40 heads, L = 48 832) on the PRODUCT library, a few launches: target of the PMC
passes of tools/run_attn_pmc.sh.  argv[1]: "prescaled" (default: queries in log2 units -> scail_attn4_m16f) or "raw" (scail_attn4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "prescaled"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, H, L = 2, 40, 48832
D = H * 128
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, L, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
if mode == "prescaled":
    q.copy_((q.float() * ops.ATTN_LOG2_SCALE).to(torch.bfloat16))
vt = ops.transpose_v(v, H)
out = torch.empty(B, L, D, device="cuda", dtype=torch.bfloat16)
for _ in range(n):
    ops.flash_attn(q, k, vt, out=out, q_prescaled=(mode == "prescaled"))
torch.cuda.synchronize()

#!/usr/bin/env python
"""Run the vendor library (hipBLASLt through torch F.linear) on the four per-token GEMM shapes of config 2, so that a
rocprofv3 --kernel-trace of this script names the kernels (macro tile, MFMA shape, staging scheme are encoded in the names)
our q8 kernel is compared with.  Prints one JSON line per shape with the vendor's TFLOP/s."""
import json

import torch

M = 97664
for (N, K) in ((15360, 5120), (5120, 5120), (13824, 5120), (5120, 13824)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    for _ in range(2):
        y = torch.nn.functional.linear(x, w)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); y = torch.nn.functional.linear(x, w); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[2]
    print(json.dumps({"shape": [M, N, K], "vendor_ms": ms, "vendor_TFLOPs": 2.0 * M * N * K / ms / 1e9}), flush=True)
    del x, w, y

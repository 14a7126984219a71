#!/usr/bin/env python
"""The VAE leg of bench.py alone (BASELINE config 4: Wan2.1 VAE encode + decode at 81 x 512 x 896): the process rocprofv3 --kernel-trace wraps
for profiles/*_vae_kernel_stats.md.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.cuda.set_device(0)
print(json.dumps(bench.vae_leg(torch.device("cuda:0"), pmc=False)))

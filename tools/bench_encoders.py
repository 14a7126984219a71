"""Request-time conditioning encoders at their real sizes (random-init weights): UMT5-XXL text encoder (24 layers, 4096 d,
64 heads, 512 tokens) and the CLIP ViT-H/14 visual tower to block 31 (257 tokens).  They run once per request."""
import json
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scail_amd.umt5 import T5EncoderModel
from scail_amd.clip import CLIPModel


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


t5 = T5EncoderModel(device="cuda")
ids = torch.randint(0, 1000, (1, 512), device="cuda")
mask = torch.zeros(1, 512, dtype=torch.int64, device="cuda"); mask[:, :64] = 1
ms = timeit(lambda: t5(ids, mask))
out = t5(ids, mask)
print(json.dumps(dict(case="UMT5-XXL encoder, 512 tokens", ms=ms, out=list(out.shape), finite=bool(torch.isfinite(out.float()).all()))))
del t5
clip = CLIPModel(device="cuda")
img = (torch.rand(3, 1, 512, 896, device="cuda") * 2 - 1)
ms = timeit(lambda: clip.visual([img]))
out = clip.visual([img])
print(json.dumps(dict(case="CLIP ViT-H/14 visual, 1 frame", ms=ms, out=list(out.shape), finite=bool(torch.isfinite(out.float()).all()))))

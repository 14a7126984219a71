"""Pin oracle/encoders_oracle.py to outputs of the REAL reference T5Encoder / VisionTransformer."""
import os

import numpy as np
import torch

from oracle import encoders_oracle as E

T5 = dict(num_heads=2, num_layers=2, num_buckets=32)
VIT = dict(num_heads=2, num_layers=3, patch=14)


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "encoders_tiny.npz"))
    out = {}
    for k in z.files:
        a = torch.from_numpy(np.asarray(z[k]))
        out[k] = a.view(torch.bfloat16).float() if a.dtype == torch.int16 else a
    return out


def split(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def test_t5_encoder_matches_reference(golden_dir):
    g = load(golden_dir)
    out = E.t5_encoder(split(g, "t5."), g["ids"], g["mask"], **T5)
    torch.testing.assert_close(out, g["t5_out"], rtol=1e-4, atol=1e-4)


def test_clip_visual_matches_reference(golden_dir):
    g = load(golden_dir)
    out = E.clip_visual_31(split(g, "vit."), g["imgs"], **VIT)
    torch.testing.assert_close(out, g["vit_out"], rtol=1e-4, atol=1e-4)

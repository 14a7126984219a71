"""tests/golden/encoders_tiny.npz from the REAL reference T5Encoder / VisionTransformer (CPU fp32).
Build-container only.  Usage: python oracle/gen_golden_encoders.py"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shims  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
T5 = dict(vocab=100, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=2, num_buckets=32)
VIT = dict(image_size=56, patch_size=14, dim=192, mlp_ratio=4, out_dim=64, num_heads=2, num_layers=3)


def bfr(sd):
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}


def main():
    torch.cuda.current_device = lambda: "cpu"           # default-arg evaluated at import (umt5.py:480, clip.py:492)
    if "ftfy" not in sys.modules:
        sys.modules["ftfy"] = MagicMock()
    ref_shims.load_reference()
    from sgm.modules.encoders import clip, umt5
    torch.manual_seed(11)
    enc = umt5.T5Encoder(shared_pos=False, dropout=0.0, **T5).eval()
    enc.load_state_dict(bfr(enc.state_dict()))
    ids = torch.randint(0, 100, (2, 40))
    mask = torch.ones(2, 40, dtype=torch.long)
    mask[1, 17:] = 0
    with torch.no_grad():
        t5_out = enc(ids, mask)
    vit = clip.VisionTransformer(pool_type="token", pre_norm=True, post_norm=False, activation="gelu", **VIT).eval()
    vit.load_state_dict(bfr(vit.state_dict()))
    imgs = torch.randn(2, 3, 56, 56).to(torch.bfloat16).float()
    with torch.no_grad():
        vit_out = vit(imgs, use_31_block=True)
    np.savez_compressed(os.path.join(OUT, "encoders_tiny.npz"), ids=ids.numpy(), mask=mask.numpy(), t5_out=t5_out.numpy(),
                        imgs=imgs.numpy(), vit_out=vit_out.numpy(),
                        # weights are bf16-exact: store the bf16 bit patterns (uint16) to halve the fixture
                        **{"t5." + k: v.to(torch.bfloat16).view(torch.int16).numpy() for k, v in enc.state_dict().items()},
                        **{"vit." + k: v.to(torch.bfloat16).view(torch.int16).numpy() for k, v in vit.state_dict().items()})
    print("t5", tuple(t5_out.shape), float(t5_out.abs().mean()), "vit", tuple(vit_out.shape), float(vit_out.abs().mean()))


if __name__ == "__main__":
    main()

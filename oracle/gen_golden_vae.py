"""Generate tests/golden/vae_tiny.npz, vae_tiny2.npz and vae_dim96.npz by running the REAL reference WanVAE_ (chunked, feature caches)
on CPU in fp32.  Build-container only.  Usage: python oracle/gen_golden_vae.py"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join("/root/reference", "sgm", "models"))

from oracle import wan_vae_oracle as V  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    import wan_vae as ref                      # sgm/models/wan_vae.py imports standalone (torch + einops)
    cfg = V.VAEConfig(dim=32, z_dim=16)
    sd = V.make_state_dict(cfg, seed=4321)
    model = ref.WanVAE_(dim=cfg.dim, z_dim=cfg.z_dim, dim_mult=list(cfg.dim_mult), num_res_blocks=2, attn_scales=[],
                        temperal_downsample=list(cfg.temperal_downsample), dropout=0.0).eval()
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    scale = [torch.tensor(V.LATENT_MEAN), 1.0 / torch.tensor(V.LATENT_STD)]
    g = torch.Generator().manual_seed(5)
    video = (torch.rand(1, 3, 9, 32, 48, generator=g) * 2 - 1).to(torch.bfloat16).float()
    z_in = torch.randn(1, 16, 3, 4, 6, generator=g).to(torch.bfloat16).float()
    with torch.no_grad():
        mu = model.encode(video, scale)
        rec = model.decode(z_in, scale).clamp(-1, 1)
        mu1 = model.encode(video[:, :, :1], scale)          # single image (the ref-frame encode of the CLI)
    np.savez_compressed(os.path.join(OUT, "vae_tiny.npz"), seed=4321, dim=32, video=video.numpy(), mu=mu.numpy(),
                        z_in=z_in.numpy(), rec=rec.numpy(), mu1=mu1.numpy())
    print("vae_tiny: mu", tuple(mu.shape), float(mu.abs().mean()), "rec", tuple(rec.shape), float(rec.abs().mean()))

    # second case: 48 base channels (48 / 96 / 192 / 192: the 96- and 192-channel kernel paths of the full model), 17 frames
    # (five streaming chunks in the reference), portrait 40 x 24, other seed
    cfg = V.VAEConfig(dim=48, z_dim=16)
    sd = V.make_state_dict(cfg, seed=99)
    model = ref.WanVAE_(dim=cfg.dim, z_dim=cfg.z_dim, dim_mult=list(cfg.dim_mult), num_res_blocks=2, attn_scales=[],
                        temperal_downsample=list(cfg.temperal_downsample), dropout=0.0).eval()
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    g = torch.Generator().manual_seed(6)
    video = (torch.rand(1, 3, 17, 40, 24, generator=g) * 2 - 1).to(torch.bfloat16).float()
    z_in = torch.randn(1, 16, 4, 5, 3, generator=g).to(torch.bfloat16).float()
    with torch.no_grad():
        mu = model.encode(video, scale)
        rec = model.decode(z_in, scale).clamp(-1, 1)
        mu1 = model.encode(video[:, :, :1], scale)
    np.savez_compressed(os.path.join(OUT, "vae_tiny2.npz"), seed=99, dim=48, video=video.numpy(), mu=mu.numpy(),
                        z_in=z_in.numpy(), rec=rec.numpy(), mu1=mu1.numpy())
    print("vae_tiny2: mu", tuple(mu.shape), float(mu.abs().mean()), "rec", tuple(rec.shape), float(rec.abs().mean()))

    # third case: the FULL model's widths (96 base channels -> 96 / 192 / 384 / 384: every kernel path the 14B pipeline's VAE
    # takes, incl. the fused conv -> RMS_norm -> SiLU epilogue that needs all 96 channels in one N tile), 9 frames, 32 x 32
    cfg = V.VAEConfig(dim=96, z_dim=16)
    sd = V.make_state_dict(cfg, seed=7)
    model = ref.WanVAE_(dim=cfg.dim, z_dim=cfg.z_dim, dim_mult=list(cfg.dim_mult), num_res_blocks=2, attn_scales=[],
                        temperal_downsample=list(cfg.temperal_downsample), dropout=0.0).eval()
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    g = torch.Generator().manual_seed(8)
    video = (torch.rand(1, 3, 9, 32, 32, generator=g) * 2 - 1).to(torch.bfloat16).float()
    z_in = torch.randn(1, 16, 3, 4, 4, generator=g).to(torch.bfloat16).float()
    with torch.no_grad():
        mu = model.encode(video, scale)
        rec = model.decode(z_in, scale).clamp(-1, 1)
        mu1 = model.encode(video[:, :, :1], scale)
    np.savez_compressed(os.path.join(OUT, "vae_dim96.npz"), seed=7, dim=96, video=video.numpy(), mu=mu.numpy(),
                        z_in=z_in.numpy(), rec=rec.numpy(), mu1=mu1.numpy())
    print("vae_dim96: mu", tuple(mu.shape), float(mu.abs().mean()), "rec", tuple(rec.shape), float(rec.abs().mean()))


if __name__ == "__main__":
    main()

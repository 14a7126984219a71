"""Checkpoint ingest on the GPU: reference-format files written from seeded weights are loaded
through scail_amd.checkpoint into the HIP weight arena and a forward is compared with the REAL reference's golden output.
  * DiT: <load>/latest + <iter>/mp_rank_0{0,1}_model_states.pt ('module' keys model.diffusion_model.*), a 2-way
    tensor-parallel pair merged on load (sat/training/model_io.py:260-327, sat/mpu/operation.py:96-124);
  * VAE: a bare state dict like Wan2.1_VAE.pth through the WanVAE(vae_pth=...) constructor (sgm/models/wan_vae.py:607-616)."""
import os

import numpy as np
import pytest
import torch

from oracle import scail_oracle as O
from oracle import wan_vae_oracle as V

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _load(golden_dir, name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, name)).items()}


@pytest.mark.parametrize("mp", [1, 2])
def test_dit_checkpoint_ingest_then_forward_equals_reference_golden(golden_dir, tmp_path, mp):
    from scail_amd import checkpoint
    from scail_amd.dit import DiffusionTransformer
    g = _load(golden_dir, "dit_tiny.npz")
    cfg = O.DiTConfig(**O.TINY)
    sd = {"model.diffusion_model." + k: v.to(torch.bfloat16) for k, v in O.make_state_dict(cfg, seed=int(g["seed"])).items()}
    sd["model.something_else.weight"] = torch.zeros(3)                       # other prefixes are filtered out
    checkpoint.save_checkpoint(sd, str(tmp_path), 1000, model_parallel_size=mp)
    assert sorted(os.listdir(tmp_path / "1000")) == [f"mp_rank_{r:02d}_model_states.pt" for r in range(mp)]
    net = DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=cfg.num_frames, latent_width=cfg.latent_width,
                               latent_height=cfg.latent_height, hidden_size=cfg.hidden_size, text_dim=cfg.text_dim,
                               num_layers=cfg.num_layers, num_attention_heads=cfg.num_attention_heads,
                               time_freq_dim=cfg.time_freq_dim, time_embed_dim=cfg.time_embed_dim, share_adaln=True,
                               inner_hidden_size=cfg.inner_hidden_size, use_i2v_clip=True, device=DEV, init_seed=99)
    kw = dict(concat_images=torch.zeros(1, device=DEV), ref_concat=g["ref"].to(DEV), concat_smpl_render=g["pose"].to(DEV),
              image_clip_features=g["clip"].to(DEV))
    before = net(g["x"].to(DEV), timesteps=g["t"].to(DEV), context=g["ctx"].to(DEV), **kw).float().cpu()
    assert (before - g["out"]).abs().max() > 0.1                             # random init: not the golden
    it, missing, unexpected = checkpoint.load_checkpoint(net, str(tmp_path), prefix="model.diffusion_model.")
    assert it == 1000 and not missing and not unexpected
    out = net(g["x"].to(DEV), timesteps=g["t"].to(DEV), context=g["ctx"].to(DEV), **kw).float().cpu()
    torch.testing.assert_close(out, g["out"], rtol=2e-2, atol=2e-2)          # the reference's own output


def test_vae_pth_ingest_then_encode_decode_equals_reference_golden(golden_dir, tmp_path):
    from scail_amd.wan_vae import WanVAE, LATENT_MEAN, LATENT_STD
    g = _load(golden_dir, "vae_tiny.npz")
    cfg = V.VAEConfig(dim=int(g["dim"]), z_dim=16)
    pth = str(tmp_path / "Wan2.1_VAE.pth")
    torch.save(V.make_state_dict(cfg, seed=int(g["seed"])), pth)
    vae = WanVAE(z_dim=16, vae_pth=pth, dtype="torch.bfloat16", device=DEV, dim=cfg.dim)
    mu = vae.model.encode(g["video"].to(DEV)).cpu()
    torch.testing.assert_close(mu, g["mu"], rtol=3e-2, atol=3e-2)
    rec = vae.model.decode(g["z_in"].to(DEV)).clamp(-1, 1).cpu()
    torch.testing.assert_close(rec, g["rec"], rtol=3e-2, atol=3e-2)
    with pytest.raises(FileNotFoundError):
        WanVAE(z_dim=16, vae_pth=str(tmp_path / "missing.pth"), device=DEV, dim=cfg.dim)

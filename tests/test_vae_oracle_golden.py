"""Pin the whole-sequence VAE oracle (oracle/wan_vae_oracle.py) to outputs of the REAL chunked
reference WanVAE_ (tests/golden/vae_tiny.npz, vae_tiny2.npz; oracle/gen_golden_vae.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import wan_vae_oracle as V

FIXTURES = ["vae_tiny.npz", "vae_tiny2.npz", "vae_dim96.npz"]      # dim 32, 9 frames 32x48; dim 48 (96 / 192-channel stages), 17 frames 40x24; dim 96 = the full model width (96 / 192 / 384), 9 frames 32x32


def _load(golden_dir, name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def test_vae_state_dict_param_count():
    cfg = V.VAEConfig(dim=96, z_dim=16)
    n = sum(int(np.prod(s)) for s in V.state_dict_spec(cfg).values())
    assert abs(n - 126.9e6) < 0.2e6           # 126.9 M parameters


@pytest.mark.parametrize("name", FIXTURES)
def test_encode_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    cfg = V.VAEConfig(dim=int(g["dim"]), z_dim=16)
    sd = V.make_state_dict(cfg, seed=int(g["seed"]))
    with torch.no_grad():
        mu = V.encode(cfg, sd, g["video"])
        mu1 = V.encode(cfg, sd, g["video"][:, :, :1])
    torch.testing.assert_close(mu, g["mu"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(mu1, g["mu1"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", FIXTURES)
def test_decode_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    cfg = V.VAEConfig(dim=int(g["dim"]), z_dim=16)
    sd = V.make_state_dict(cfg, seed=int(g["seed"]))
    with torch.no_grad():
        rec = V.decode(cfg, sd, g["z_in"])
    torch.testing.assert_close(rec, g["rec"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["vae_tiny.npz", "vae_tiny2.npz"])
def test_tap_matmul_convolutions_match_reference(golden_dir, name, monkeypatch):
    """CONV_IMPL = "taps" (every convolution as one fp32 matmul per kernel tap over a zero-padded channels-last copy: what lets the oracle
    run in fp32 on the GPU at config 4's own size, tests/test_vae_gpu.py::test_vae_fullsize_vs_fp32) is the same function: pinned to the
    real chunked reference's golden like the F.conv3d form, including the stride-2 / upsample / temporal convolutions."""
    g = _load(golden_dir, name)
    cfg = V.VAEConfig(dim=int(g["dim"]), z_dim=16)
    sd = V.make_state_dict(cfg, seed=int(g["seed"]))
    monkeypatch.setattr(V, "CONV_IMPL", "taps")
    with torch.no_grad():
        mu = V.encode(cfg, sd, g["video"])
        rec = V.decode(cfg, sd, g["z_in"])
    torch.testing.assert_close(mu, g["mu"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rec, g["rec"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name,run", [("vae_tiny2.npz", 1), ("vae_tiny2.npz", 2), ("vae_dim96.npz", 2), ("vae_tiny.npz", 5)])
@pytest.mark.parametrize("impl", ["conv", "taps"])
def test_upsample_in_runs_of_frames_matches_reference(golden_dir, name, run, impl, monkeypatch):
    """oracle/wan_vae_oracle.py:upsample evaluates the per-frame interpolate + 3x3 convolution in RUNS of frames (round 4: torch-ROCm's
    F.interpolate mis-indexes > 2^32-element strided tensors, so config 4's own size needs several runs).  The goldens are small enough
    for a single run, which leaves the multi-run branch (slice -> permute -> concat -> reshape) unpinned -- here the run length is
    forced to 1, 2 and 5 frames (5: a ragged last run for the 17- and 9-frame stages) and decode must still equal the REAL chunked
    reference's output (wan_vae.py:57-63, 101-141), for both convolution forms the full-size GPU test uses."""
    g = _load(golden_dir, name)
    cfg = V.VAEConfig(dim=int(g["dim"]), z_dim=16)
    sd = V.make_state_dict(cfg, seed=int(g["seed"]))
    monkeypatch.setattr(V, "UPSAMPLE_RUN_FRAMES", run)
    if impl == "taps":
        monkeypatch.setattr(V, "CONV_IMPL", "taps")
    calls = []
    interp = V.F.interpolate
    monkeypatch.setattr(V.F, "interpolate", lambda *a, **k: (calls.append(1), interp(*a, **k))[1])
    with torch.no_grad():
        rec = V.decode(cfg, sd, g["z_in"])
    assert len(calls) > 3, "the multi-run branch did not run (3 upsample layers, one interpolate call per run)"
    torch.testing.assert_close(rec, g["rec"], rtol=1e-4, atol=1e-4)

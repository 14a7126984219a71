"""Pin the CPU oracle (oracle/scail_oracle.py) to outputs of the REAL reference
(tests/golden/*.npz, produced by oracle/gen_golden.py in the build container)."""
import os

import numpy as np
import torch

from oracle import scail_oracle as O


def _load(golden_dir, name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def test_state_dict_spec_matches_survey_count():
    cfg = O.DiTConfig(hidden_size=128, num_layers=2, num_attention_heads=4, inner_hidden_size=256, text_dim=64,
                      time_freq_dim=256, time_embed_dim=128)
    assert len(O.state_dict_spec(cfg)) == 73           # SURVEY.md Appendix B [probe]
    n = sum(int(np.prod(s)) for s in O.state_dict_spec(cfg).values())
    assert n == 2474432                                 # reference parameter count for the survey's probe config


import pytest


@pytest.mark.parametrize("name,cfgd", [("dit_tiny.npz", O.TINY), ("dit_config1.npz", O.CONFIG1), ("dit_shapes.npz", O.TINY)])
def test_dit_forward_matches_reference(golden_dir, name, cfgd):
    g = _load(golden_dir, name)
    cfg = O.DiTConfig(**cfgd)
    sd = O.make_state_dict(cfg, seed=int(g["seed"]))
    out, hidden = O.dit_forward(cfg, sd, g["x"], g["t"], g["ctx"], g["ref"], g["pose"], g["clip"], return_hidden=True)
    for i in range(cfg.num_layers):
        torch.testing.assert_close(hidden[i + 1], g[f"hidden{i + 1}"], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(out, g["out"], rtol=2e-5, atol=2e-5)


def test_rope_tables_match_reference(golden_dir):
    g = _load(golden_dir, "rope_tiny.npz")
    cfg = O.DiTConfig(**O.TINY)
    cos, sin = O.rope_tables(cfg, int(g["rope_T"]), int(g["rope_H"]), int(g["rope_W"]))
    torch.testing.assert_close(O.apply_rope(g["q"], cos, sin), g["qr"], rtol=1e-5, atol=2e-6)
    # interleaved pairs share one angle, pooled or not: the HIP path stores (L, hd/2) tables
    assert torch.equal(cos[:, 0::2], cos[:, 1::2]) and torch.equal(sin[:, 0::2], sin[:, 1::2])


def test_rope_tables_sequence_parallel_shift(golden_dir):
    g = _load(golden_dir, "rope_tiny_sp.npz")
    cfg = O.DiTConfig(**O.TINY)
    for r in range(2):
        cos, sin = O.rope_tables(cfg, 4, 2, 4, H_shift=r * 2)
        torch.testing.assert_close(O.apply_rope(g[f"q{r}"], cos, sin), g[f"qr{r}"], rtol=1e-5, atol=2e-6)


def test_rope_tables_sequence_parallel_w_shift(golden_dir):
    """Portrait latents are split along W (chunk_dim 4): rank-shifted W window, also inside the pooled pose tables."""
    from scail_amd import rope
    g = _load(golden_dir, "rope_tiny_sp_w.npz")
    cfg = O.DiTConfig(**O.TINY)
    for r in range(2):
        cos, sin = O.rope_tables(cfg, 2, 8, 6, W_shift=r * 6)
        torch.testing.assert_close(O.apply_rope(g[f"q{r}"], cos, sin), g[f"qr{r}"], rtol=1e-5, atol=2e-6)
        ch, sh = rope.build_tables(cfg.head_dim, 2, 8, 6, W_shift=r * 6)              # the host tables the HIP kernel consumes
        assert torch.equal(cos[:, 0::2], ch) and torch.equal(sin[:, 0::2], sh)


def test_sampler_matches_reference(golden_dir):
    g = _load(golden_dir, "sampler_tiny.npz")
    d = _load(golden_dir, "dit_tiny.npz")
    cfg = O.DiTConfig(**O.TINY)
    sd = O.make_state_dict(cfg, seed=int(d["seed"]))
    torch.testing.assert_close(O.flow_sigmas(2), g["sigmas"], rtol=0, atol=0)
    xT, _ = O.sample(cfg, sd, g["x0"], g["c_ctx"], g["uc_ctx"], d["ref"], d["pose"], d["clip"], num_steps=2)
    torch.testing.assert_close(xT, g["xT"], rtol=5e-5, atol=5e-5)


def test_sampler_50_steps_matches_reference(golden_dir):
    """The shipped step count (50, shift 5, CFG 4) on BASELINE config 1's network: the restatement follows the real
    reference's trajectory (latents after 2 / 10 / 25 / 50 steps), i.e. the oracle is pinned over the whole schedule and
    not just its first two steps."""
    g = _load(golden_dir, "sampler_tiny_50.npz")
    cfg = O.DiTConfig(**O.CONFIG1)
    sd = O.make_state_dict(cfg, seed=int(g["seed"]))
    xT, traj = O.sample(cfg, sd, g["x0"], g["c_ctx"], g["uc_ctx"], g["ref"], g["pose"], g["clip"], num_steps=50)
    assert len(traj) == 50
    for k in (2, 10, 25):
        torch.testing.assert_close(traj[k - 1], g[f"x{k}"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xT, g["xT"], rtol=2e-4, atol=2e-4)


def test_request_pipeline_composition_matches_reference(golden_dir):
    """e2e_tiny.npz (oracle/gen_golden_e2e.py): the reference's own encode_first_stage -> sample -> decode_first_stage chain with the
    script lines of sample_video.py between them.  The restatements composed the same way (pixels -> half-resolution pose ->
    VAE mean x scale factor -> b c t h w <-> b t c h w -> 3 CFG / Euler steps on the global RNG stream's noise -> 1 / scale
    factor -> decode -> clamp((x + 1) / 2)) reproduce every stage -- the checker of tests/test_dit_gpu.py's pipeline test."""
    import torch.nn.functional as F
    from oracle import wan_vae_oracle as V
    g = _load(golden_dir, "e2e_tiny.npz")
    bf = lambda t: t.to(torch.bfloat16).float()
    sf = float(g["scale_factor"])
    vcfg = V.VAEConfig(dim=32, z_dim=16)
    vsd = V.make_state_dict(vcfg, seed=int(g["vae_seed"]))
    cfg = O.DiTConfig(**O.CONFIG1)
    sd = O.make_state_dict(cfg, seed=int(g["dit_seed"]))
    ref_px = bf((g["ref_u8"].float() / 255.0) * 2 - 1)                                        # (1, 3, H, W)
    pose_px = (g["pose_u8"].permute(0, 3, 1, 2).float() - 127.5) / 127.5
    smpl_px = bf(F.interpolate(pose_px, scale_factor=0.5, mode="bilinear", align_corners=False))
    ref_concat = (sf * V.encode(vcfg, vsd, ref_px.unsqueeze(2))).permute(0, 2, 1, 3, 4)      # B C 1 H W -> B 1 C H W
    smpl_lat = (sf * V.encode(vcfg, vsd, smpl_px.permute(1, 0, 2, 3).unsqueeze(0))).permute(0, 2, 1, 3, 4)
    torch.testing.assert_close(ref_concat, g["ref_concat"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(smpl_lat, g["smpl_render_latent"], rtol=1e-4, atol=1e-4)
    torch.manual_seed(int(g["noise_seed"]))
    x0 = torch.randn(1, smpl_lat.shape[1], 16, ref_concat.shape[3], ref_concat.shape[4])
    z, _ = O.sample(cfg, sd, x0, g["ctx"], g["uc_ctx"], ref_concat, smpl_lat, g["clip"], num_steps=int(g["steps"]))
    z = z.permute(0, 2, 1, 3, 4)
    torch.testing.assert_close(z, g["samples_z"], rtol=2e-4, atol=2e-4)
    video = torch.clamp((V.decode(vcfg, vsd, z / sf).permute(0, 2, 1, 3, 4) + 1.0) / 2.0, 0.0, 1.0)
    torch.testing.assert_close(video, g["samples"].float(), rtol=0, atol=1e-3)             # the fixture stores the video as fp16


def test_sampler_long_matches_reference(golden_dir):
    """RFSamplerLong (temporal tiling, sampling.py:986-1085) on the real reference vs the restatement."""
    g = _load(golden_dir, "sampler_long_tiny.npz")
    d = _load(golden_dir, "dit_tiny.npz")
    cfg = O.DiTConfig(**O.TINY)
    sd = O.make_state_dict(cfg, seed=int(d["seed"]))
    tiles = [list(map(int, r)) for r in g["tiles"]]
    xT = O.sample_long(cfg, sd, g["x0"], g["c_ctx"], g["uc_ctx"], d["ref"], g["smpl_tiled"], d["clip"], tiles, num_steps=2)
    torch.testing.assert_close(xT, g["xT"], rtol=5e-5, atol=5e-5)
    w = O.tile_weight(4)
    assert torch.allclose(w, torch.tensor([0.25, 0.75, 0.75, 0.25]))
    with pytest.raises(ValueError):
        O.sample_long(cfg, sd, g["x0"][:, :4], g["c_ctx"], g["uc_ctx"], d["ref"], g["smpl_tiled"][:, :1], d["clip"],
                      tiles[:1], num_steps=1)


def test_sigmas50(golden_dir):
    g = _load(golden_dir, "sigmas50.npz")
    s = O.flow_sigmas(50)
    assert torch.equal(s, g["sigmas"])
    assert s[0] == 1.0 and s[-1] == 0.0 and abs(float(s[1]) - 0.9959) < 1e-4


def test_multi_character_rope_extension_is_consistent():
    """BASELINE config 5 extension (not in the reference): one character == the pinned tables; the host tables
    (scail_amd/rope.py) equal the oracle's; the second character's windows do not collide with any other segment's."""
    from scail_amd import rope
    cfg = O.DiTConfig(**O.TINY)
    c1, s1 = O.rope_tables(cfg, 4, 4, 4)
    c1m, s1m = O.rope_tables_multi(cfg, 4, 4, 4, 1)
    assert torch.equal(c1, c1m) and torch.equal(s1, s1m)
    c2, s2 = O.rope_tables_multi(cfg, 4, 4, 4, 2, H_shift=2)
    ch, sh = rope.build_tables(cfg.head_dim, 4, 4, 4, H_shift=2, n_char=2)
    assert torch.equal(c2[:, 0::2], ch) and torch.equal(s2[:, 0::2], sh)
    lref, lnoise, lpose = 16, 64, 16
    assert c2.shape[0] == 2 * lref + lnoise + 2 * lpose
    # character 0 and the noise tokens keep the reference's positions
    c0, s0 = O.rope_tables(cfg, 4, 4, 4, H_shift=2)
    assert torch.equal(c2[:lref], c0[:lref]) and torch.equal(c2[2 * lref:2 * lref + lnoise], c0[lref:lref + lnoise])
    assert torch.equal(c2[2 * lref + lnoise:2 * lref + lnoise + lpose], c0[lref + lnoise:])
    # no two tokens share a position (rows of [cos | sin] are pairwise distinct)
    rows = torch.cat([c2, s2], 1)
    assert torch.unique(rows, dim=0).shape[0] == rows.shape[0]
    with pytest.raises(ValueError, match="table extent"):
        rope.build_tables(cfg.head_dim, 4, 4, 4, n_char=3, max_W=cfg.latent_width // 2 + 120)

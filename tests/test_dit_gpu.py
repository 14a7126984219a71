"""GPU parity tests, network level: the HIP DiT behind the reference's network interface against
(a) the golden outputs of the REAL reference (tests/golden, fp32 CPU) and (b) the CPU oracle's
per-block hidden states, plus the sampler loop.  Tolerance: bf16 storage / fp32 accumulate vs the
fp32 reference -> rtol 2e-2, atol 2e-2 per block output; cosine >= 0.999 on final latents
(BASELINE.md section 3)."""
import os

import numpy as np
import pytest
import torch

from oracle import scail_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _load(golden_dir, name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _net(cfgd, seed):
    from scail_amd.dit import DiffusionTransformer
    cfg = O.DiTConfig(**cfgd)
    net = DiffusionTransformer(
        transformer_args=dict(model_parallel_size=1, is_decoder=True), num_frames=cfg.num_frames,
        time_compressed_rate=4, latent_width=cfg.latent_width, latent_height=cfg.latent_height,
        patch_size=[1, 2, 2], in_channels=20, out_channels=16, hidden_size=cfg.hidden_size, text_dim=cfg.text_dim,
        num_layers=cfg.num_layers, num_attention_heads=cfg.num_attention_heads, elementwise_affine=False,
        time_freq_dim=cfg.time_freq_dim, time_embed_dim=cfg.time_embed_dim, share_adaln=True,
        inner_hidden_size=cfg.inner_hidden_size, use_i2v_clip=True, dtype="bf16", device=DEV,
        modules={"pos_embed_config": {"params": {"hidden_size_head": 128, "interleaved_rope": True}},
                 "adaln_layer_config": {"params": {"qk_ln": True, "hidden_size_head": cfg.hidden_size}}})
    sd = O.make_state_dict(cfg, seed=seed)
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return cfg, sd, net


def _cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a @ b) / (a.norm() * b.norm()))


@pytest.mark.parametrize("name,cfgd", [("dit_tiny.npz", O.TINY), ("dit_config1.npz", O.CONFIG1), ("dit_shapes.npz", O.TINY)])
def test_dit_forward_vs_reference_golden(golden_dir, name, cfgd):
    g = _load(golden_dir, name)
    cfg, sd, net = _net(cfgd, int(g["seed"]))
    hidden = {}
    net._tap = lambda i, h: hidden.__setitem__(i, h.float().cpu().clone())
    out = net(g["x"].to(DEV), timesteps=g["t"].to(DEV), context=g["ctx"].to(DEV),
              concat_images=torch.zeros(1, *g["x"].shape[1:], device=DEV), ref_concat=g["ref"].to(DEV),
              concat_smpl_render=g["pose"].to(DEV), image_clip_features=g["clip"].to(DEV))
    assert out.dtype == torch.bfloat16 and out.shape == g["out"].shape
    # embedding output vs oracle, block outputs vs the reference's own hidden states
    _, oh = O.dit_forward(cfg, sd, g["x"], g["t"], g["ctx"], g["ref"], g["pose"], g["clip"], return_hidden=True)
    torch.testing.assert_close(hidden[-1], oh[0], rtol=2e-2, atol=2e-2)
    for i in range(cfg.num_layers):
        torch.testing.assert_close(hidden[i], g[f"hidden{i + 1}"], rtol=2e-2, atol=2e-2, msg=lambda m: f"block {i}: {m}")
    torch.testing.assert_close(out.float().cpu(), g["out"], rtol=2e-2, atol=2e-2)
    assert _cos(out.float().cpu(), g["out"]) >= 0.999


@pytest.mark.parametrize("name,cfgd", [("dit_tiny.npz", O.TINY), ("dit_config1.npz", O.CONFIG1), ("dit_shapes.npz", O.TINY)])
def test_c_step_executor_equals_host_orchestration_and_golden(golden_dir, name, cfgd):
    """include/scail_dit.h: the whole evaluation as one C call.  It enqueues the same kernels in the same order as the
    Python orchestration, so the two must agree BIT FOR BIT; and it must match the reference golden like the other path.
    A captured hipGraph of the step must replay to the same result (no host sync inside the call)."""
    g = _load(golden_dir, name)
    cfg, sd, net = _net(cfgd, int(g["seed"]))
    kw = dict(concat_images=torch.zeros(1, *g["x"].shape[1:], device=DEV), ref_concat=g["ref"].to(DEV),
              concat_smpl_render=g["pose"].to(DEV), image_clip_features=g["clip"].to(DEV))
    x, t, ctx = g["x"].to(DEV), g["t"].to(DEV), g["ctx"].to(DEV)
    net.use_c_step = False
    o_py = net.forward_f32(x, t, ctx, None, **kw)
    net.use_c_step = True
    o_c = net.forward_f32(x, t, ctx, None, **kw)
    assert net._cstep is not None, "the C executor was not used"
    assert torch.equal(o_c, o_py)
    torch.testing.assert_close(o_c.cpu(), g["out"], rtol=2e-2, atol=2e-2)
    # stream capture -> graph replay
    # (cond_key: the conditioning cache is then matched by key, not by a device-side tensor comparison that would sync)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        net.forward_f32(x, t, ctx, None, cond_key="graph", **kw)   # warm-up on the capture stream (lazy attribute setup)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        o_g = net.forward_f32(x, t, ctx, None, cond_key="graph", **kw)
    o_g.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(o_g, o_py)


def test_c_block_entry_point(golden_dir):
    """Seam B2 (scail_dit_block): feeding the per-op path's hidden states after the embedding through the C block entry
    point layer by layer reproduces the per-op path's hidden states bit for bit (and therefore the reference's, which
    test_dit_forward_vs_reference_golden pins)."""
    from scail_amd import ops
    from scail_amd.cstep import CStep
    g = _load(golden_dir, "dit_tiny.npz")
    cfg, sd, net = _net(O.TINY, int(g["seed"]))
    hidden = {}
    net._tap = lambda i, h: hidden.__setitem__(i, h.clone())
    kw = dict(concat_images=torch.zeros(1, *g["x"].shape[1:], device=DEV), ref_concat=g["ref"].to(DEV),
              concat_smpl_render=g["pose"].to(DEV), image_clip_features=g["clip"].to(DEV))
    net.forward_f32(g["x"].to(DEV), g["t"].to(DEV), g["ctx"].to(DEV), None, **kw)
    net._tap = None
    W = net.prepare()
    cs = CStep(net, W)
    D = net.hidden_size
    # the layer's modulation rows, exactly as the host path builds them (time embedding -> adaLN projection + table)
    t32 = g["t"].to(DEV).float()
    temb = ops.timestep_embedding(t32, net.time_freq_dim)
    from scail_amd import lib as L
    e1 = ops.small_linear(temb, W["time_embed.0.w"], W["time_embed.0.b"], act_out=L.ACT_SILU)
    emb = ops.small_linear(e1, W["time_embed.2.w"], W["time_embed.2.b"])
    adaln = ops.small_linear(emb, W["adaln_projection.1.w"], W["adaln_projection.1.b"], act_in=L.ACT_SILU)
    mod = ops.adaln_table(adaln, W["adaln_tables"])
    cond = net._cond_cache
    T, H, Wd = g["x"].shape[1], g["x"].shape[3], g["x"].shape[4]
    cos, sin = net._rope(T, H // 2, Wd // 2, 0, 0, torch.device(DEV))
    h = hidden[-1].clone()
    for i in range(cfg.num_layers):
        cs.block(i, h, mod[i].contiguous(), cond, cos, sin)
        assert torch.equal(h, hidden[i]), f"block {i}"


def test_full_width_layer_vs_oracle():
    """BASELINE config-2 WIDTH (D = 5120, 40 heads of 128, FF = 13824, text 4096) on a short sequence: one transformer layer
    + embeddings + final layer against the fp32 oracle.  Exercises the shapes the 14B step uses (40-head attention, the
    256-wide GEMM tiles on N = 15360 / 13824 / 5120, 5120-wide norms) where the goldens use 256-wide toy networks."""
    cfgd = dict(hidden_size=5120, num_layers=1, num_attention_heads=40, inner_hidden_size=13824, text_dim=4096,
                time_freq_dim=256, time_embed_dim=5120, latent_height=64, latent_width=64, num_frames=13)
    cfg, sd, net = _net(cfgd, 4321)
    g = torch.Generator().manual_seed(5)
    r = lambda *sh: torch.randn(*sh, generator=g).to(torch.bfloat16).float()
    T, H, W = 3, 16, 24                                           # L = 96 + 288 + 72 = 456 tokens
    x, ctx = r(2, T, 16, H, W), r(2, 20, 4096)
    ctx[:, 10:] = 0
    ref, pose, clip = r(1, 1, 16, H, W), r(1, T, 16, H // 2, W // 2), r(1, 9, 1280)
    t = torch.tensor([640.0, 640.0])
    want = O.dit_forward(cfg, sd, x, t, ctx, ref, pose, clip)
    for use_c in (False, True):
        net.use_c_step = use_c
        got = net.forward_f32(x.to(DEV), t.to(DEV), ctx.to(DEV), None, concat_images=torch.zeros(1, device=DEV),
                              ref_concat=ref.to(DEV), concat_smpl_render=pose.to(DEV), image_clip_features=clip.to(DEV))
        torch.testing.assert_close(got.cpu(), want, rtol=3e-2, atol=3e-2)
        assert _cos(got.cpu(), want) >= 0.999


def test_dit_conditioning_cache_and_batch_of_one(golden_dir):
    """Same inputs twice (cache hit) and a changed prompt (cache miss) must both be right."""
    g = _load(golden_dir, "dit_tiny.npz")
    cfg, sd, net = _net(O.TINY, int(g["seed"]))
    kw = dict(concat_images=torch.zeros(1, *g["x"].shape[1:], device=DEV), ref_concat=g["ref"].to(DEV),
              concat_smpl_render=g["pose"].to(DEV), image_clip_features=g["clip"].to(DEV))
    o1 = net(g["x"].to(DEV), timesteps=g["t"].to(DEV), context=g["ctx"].to(DEV), **kw)
    o2 = net(g["x"].to(DEV), timesteps=g["t"].to(DEV), context=g["ctx"].to(DEV), **kw)
    assert torch.equal(o1, o2)
    ctx2 = g["ctx"].flip(0).contiguous()
    o3 = net(g["x"].to(DEV), timesteps=g["t"].to(DEV), context=ctx2.to(DEV), **kw)
    want = O.dit_forward(cfg, sd, g["x"], g["t"], ctx2, g["ref"], g["pose"], g["clip"])
    torch.testing.assert_close(o3.float().cpu(), want, rtol=2e-2, atol=2e-2)
    assert not torch.equal(o1, o3)


def test_cfg_pair_flag_is_checked_against_the_inputs(golden_dir):
    """SCAIL_DIT_CFG_PAIR is a statement about the inputs (x[1] == x[0], t[1] == t[0]: element 1 receives element 0's layer-0
    self-attention).  The binding verifies it the first time a network sees the key, so a guider / denoiser that sets it on a batch
    that is NOT one latent twice fails loudly instead of silently computing something else."""
    g = _load(golden_dir, "dit_tiny.npz")
    cfg, sd, net = _net(O.TINY, int(g["seed"]))
    kw = dict(concat_images=torch.zeros(1, device=DEV), ref_concat=g["ref"].to(DEV), concat_smpl_render=g["pose"].to(DEV),
              image_clip_features=g["clip"].to(DEV))
    x, t = g["x"].to(DEV), g["t"].to(DEV)
    assert not torch.equal(x[0], x[1])                                    # the golden's batch holds two different latents
    with pytest.raises(ValueError, match="cfg_pair"):
        net(x, timesteps=t, context=g["ctx"].to(DEV), cfg_pair=True, **kw)
    xx = torch.cat([x[:1], x[:1]])
    with pytest.raises(ValueError, match="cfg_pair"):
        net(xx, timesteps=torch.tensor([700.0, 701.0], device=DEV), context=g["ctx"].to(DEV), cfg_pair=True, **kw)
    a = net(xx, timesteps=t, context=g["ctx"].to(DEV), cfg_pair=True, **kw)          # a true pair: accepted, and bit-identical to the plain evaluation
    b = net(xx, timesteps=t, context=g["ctx"].to(DEV), **kw)
    assert torch.equal(a, b)


def test_sampler_two_steps_vs_reference_golden(golden_dir):
    """RFSampler + Denoiser + VanillaCFG + OpenAIWrapper protocol (generic path) and the fused HIP
    path against the reference's 2-step run (sampler_tiny.npz)."""
    from scail_amd import sampler as S
    g = _load(golden_dir, "sampler_tiny.npz")
    d = _load(golden_dir, "dit_tiny.npz")
    cfg, sd, net = _net(O.TINY, int(d["seed"]))
    smp = S.RFSampler(hunyuan_schedule=True, shift_scale=5, num_steps=2,
                      guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})
    assert torch.equal(smp.sigmas(), g["sigmas"])
    shared = dict(concat_images=torch.zeros(1, *d["x"].shape[1:], device=DEV), ref_concat=d["ref"].to(DEV),
                  concat_smpl_render=d["pose"].to(DEV), image_clip_features=d["clip"].to(DEV))
    c = dict(crossattn=g["c_ctx"].to(DEV), **shared)
    uc = dict(crossattn=g["uc_ctx"].to(DEV), **shared)
    xT = smp.sample_hip(net, g["x0"].to(DEV), c, uc)               # one C call for the whole loop (scail_dit_sample)
    net.use_c_step = False
    xT_py = smp.sample_hip(net, g["x0"].to(DEV), c, uc)            # per-step host loop over the per-op path
    net.use_c_step = True
    assert torch.equal(xT, xT_py)
    # per-forward tolerance 2e-2 (bf16) is amplified by CFG: v = v_u + 4 (v_c - v_u) carries up to
    # (2*4 - 1) = 7x the forward error, times sum |dsigma| = 1 over the run -> atol 0.14; the primary
    # criteria are the cosine and the mean error.
    torch.testing.assert_close(xT.cpu(), g["xT"], rtol=3e-2, atol=0.14)
    assert _cos(xT.cpu(), g["xT"]) >= 0.999
    assert float((xT.cpu() - g["xT"]).abs().mean()) < 1.5e-2
    den = S.Denoiser()
    wrapped = S.OpenAIWrapper(net, dtype=torch.bfloat16)
    fn = lambda inp, sigma, cc, **kw: den(wrapped, inp, sigma, cc, concat_images=None, chunk_dim=None, **kw)
    xT2 = smp(fn, g["x0"].to(DEV).clone(), dict(c), uc=dict(uc))
    torch.testing.assert_close(xT2.cpu(), xT.cpu(), rtol=1e-2, atol=1e-2)


def test_sampler_fifty_steps_drift_vs_reference_golden(golden_dir, capsys):
    """The step count config 2 actually runs (yaml :113-131: 50 steps, shift 5, CFG 4; sampling.py:965-982) on BASELINE
    config 1's network: the bf16 HIP network against the fp32 reference's own trajectory (sampler_tiny_50.npz).  The fused
    C loop (scail_dit_sample) and the per-step host loop give the same bits; the host loop's callback gives the latents after
    2 / 10 / 25 / 50 steps, so the DRIFT CURVE is on record (printed; the last measured one is in DESIGN.md section 2).
    Criterion: cosine >= 0.999 on the final latent; element bound stated below."""
    from scail_amd import sampler as S
    g = _load(golden_dir, "sampler_tiny_50.npz")
    cfg, sd, net = _net(O.CONFIG1, int(g["seed"]))
    smp = S.RFSampler(hunyuan_schedule=True, shift_scale=5, num_steps=50,
                      guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})
    shared = dict(concat_images=torch.zeros(1, *g["x0"].shape[1:], device=DEV), ref_concat=g["ref"].to(DEV),
                  concat_smpl_render=g["pose"].to(DEV), image_clip_features=g["clip"].to(DEV))
    c = dict(crossattn=g["c_ctx"].to(DEV), **shared)
    uc = dict(crossattn=g["uc_ctx"].to(DEV), **shared)
    xT = smp.sample_hip(net, g["x0"].to(DEV), c, uc)                    # scail_dit_sample: 50 steps in one C call
    traj = {}
    xT_host = smp.sample_hip(net, g["x0"].to(DEV), c, uc, step_callback=lambda i, x: traj.__setitem__(i + 1, x.cpu().clone()))
    assert torch.equal(xT, xT_host)
    curve = []
    for k in (2, 10, 25, 50):
        want = g["xT"] if k == 50 else g[f"x{k}"]
        d = (traj[k] - want).abs()
        curve.append((k, float(d.max()), float(d.mean()), _cos(traj[k], want)))
    with capsys.disabled():
        print("\n50-step drift vs the fp32 reference (step, max |d|, mean |d|, cosine):")
        for row in curve:
            print("   step %2d  max %.4f  mean %.5f  cos %.6f" % row)
    assert min(r[3] for r in curve) >= 0.999
    # element bound: the 2-step tests' 0.14 is 7 x 2e-2 x sum |dsigma| (CFG carries (2 * 4 - 1) x the forward error) -- a worst
    # case; over the whole schedule sum |dsigma| is again 1 and the per-step errors are not of one sign.  Measured (round 6, MI355X):
    # step 2 max 0.0005 / mean 0.00009, step 10 0.0020 / 0.00032, step 25 0.0052 / 0.00080, step 50 0.0199 / 0.00359, cosine
    # 0.999995 -- the error grows with 1 / sigma-like steepness near the end of the schedule but stays 7x inside the bound.
    torch.testing.assert_close(xT.cpu(), g["xT"], rtol=3e-2, atol=0.05)
    assert curve[-1][2] < 8e-3 and curve[0][1] < 5e-3
    # the reference's protocol path (Denoiser + OpenAIWrapper + VanillaCFG objects) over the same 50 steps
    den = S.Denoiser()
    wrapped = S.OpenAIWrapper(net, dtype=torch.bfloat16)
    fn = lambda inp, sigma, cc, **kw: den(wrapped, inp, sigma, cc, concat_images=None, chunk_dim=None, **kw)
    xT2 = smp(fn, g["x0"].to(DEV).clone(), dict(c), uc=dict(uc))
    assert _cos(xT2.cpu(), g["xT"]) >= 0.999
    torch.testing.assert_close(xT2.cpu(), g["xT"], rtol=3e-2, atol=0.14)


def test_sampler_long_vs_reference_golden(golden_dir):
    """RFSamplerLong (temporal tiling, sampling.py:986-1085): fused HIP path and the generic protocol against the
    reference's own 2-step run on a 6-frame latent with three overlapping 4-frame tiles."""
    from scail_amd import sampler as S
    g = _load(golden_dir, "sampler_long_tiny.npz")
    d = _load(golden_dir, "dit_tiny.npz")
    cfg, sd, net = _net(O.TINY, int(d["seed"]))
    smp = S.RFSamplerLong(hunyuan_schedule=True, shift_scale=5, num_steps=2,
                          guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})
    tiles = [list(map(int, r)) for r in g["tiles"]]
    shared = dict(concat_images=torch.zeros(1, 4, 16, 8, 8, device=DEV), ref_concat=d["ref"].to(DEV),
                  smpl_tiled=g["smpl_tiled"].to(DEV), image_clip_features=d["clip"].to(DEV))
    c = dict(crossattn=g["c_ctx"].to(DEV), **shared)
    uc = dict(crossattn=g["uc_ctx"].to(DEV), **shared)
    xT = smp.sample_hip(net, g["x0"].to(DEV), c, uc, tile_indices=tiles)
    torch.testing.assert_close(xT.cpu(), g["xT"], rtol=3e-2, atol=0.14)       # same bound as the plain sampler test
    assert _cos(xT.cpu(), g["xT"]) >= 0.999
    assert float((xT.cpu() - g["xT"]).abs().mean()) < 1.5e-2
    den = S.Denoiser()
    wrapped = S.OpenAIWrapper(net, dtype=torch.bfloat16)
    fn = lambda inp, sigma, cc, **kw: den(wrapped, inp, sigma, cc, concat_images=None, chunk_dim=None, **kw)
    xT2 = smp(fn, g["x0"].to(DEV).clone(), dict(c), uc=dict(uc), tile_indices=tiles)
    torch.testing.assert_close(xT2.cpu(), xT.cpu(), rtol=1e-2, atol=1e-2)
    with pytest.raises(ValueError):
        smp.sample_hip(net, g["x0"].to(DEV)[:, :4], c, uc, tile_indices=tiles[:1])


def test_engine_sample_from_reference_yaml_shapes():
    """SATVideoDiffusionEngine.sample through the reference-style model config (targets are the
    reference's class paths, mapped by scail_amd.config)."""
    from scail_amd.engine import SATVideoDiffusionEngine
    mc = {
        "use_i2v_clip": True,
        "denoiser_config": {"target": "sgm.modules.diffusionmodules.denoiser.Denoiser", "params": {
            "weighting_config": {"target": "sgm.modules.diffusionmodules.denoiser_weighting.EpsWeighting"},
            "scaling_config": {"target": "sgm.modules.diffusionmodules.denoiser_scaling.RFScaling"}}},
        "network_config": {"target": "dit_video_crossattn_sc_xc.DiffusionTransformer", "params": dict(
            time_freq_dim=256, time_embed_dim=128, share_adaln=True, elementwise_affine=False, num_frames=13,
            time_compressed_rate=4, latent_width=32, latent_height=32, num_layers=2, patch_size=[1, 2, 2],
            in_channels=20, out_channels=16, text_dim=64, hidden_size=128, inner_hidden_size=256,
            num_attention_heads=1, use_SwiGLU=False, use_RMSNorm=False, layernorm_epsilon=1e-6,
            transformer_args=dict(model_parallel_size=1, is_decoder=True),
            modules={"pos_embed_config": {"params": {"hidden_size_head": 128, "interleaved_rope": True}},
                     "adaln_layer_config": {"params": {"qk_ln": True, "qk_ln_affine": True, "hidden_size_head": 128}}})},
        "sampler_config": {"target": "sgm.modules.diffusionmodules.sampling.RFSampler", "params": dict(
            mode="normal", schedule_shift=False, hunyuan_schedule=True, shift_scale=5, num_steps=2, verbose=False,
            discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.RFDiscretization", "params": {"reverse": False}},
            guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})},
    }
    eng = SATVideoDiffusionEngine(mc, device=DEV)
    T, H, W = 4, 8, 8
    g = torch.Generator().manual_seed(0)
    shared = dict(concat_images=torch.zeros(1, T, 16, H, W, device=DEV),
                  ref_concat=torch.randn(1, 1, 16, H, W, generator=g).to(DEV).to(torch.bfloat16),
                  concat_smpl_render=torch.randn(1, T, 16, H // 2, W // 2, generator=g).to(DEV).to(torch.bfloat16),
                  image_clip_features=torch.randn(1, 5, 1280, generator=g).to(DEV).to(torch.bfloat16))
    c = dict(crossattn=torch.randn(1, 12, 64, generator=g).to(DEV), **shared)
    uc = dict(crossattn=torch.zeros(1, 12, 64, device=DEV), **shared)
    # seeded reference-format weights into the engine's network, so the whole sampler can be checked against the oracle's
    cfg = O.DiTConfig(**O.CONFIG1)
    sd = O.make_state_dict(cfg, seed=77)
    missing, unexpected = eng.network.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    z = eng.sample(c, uc=uc, batch_size=1, shape=(T, 16, H, W), generator=torch.Generator().manual_seed(1))
    assert z.shape == (1, T, 16, H, W) and z.dtype == torch.bfloat16 and torch.isfinite(z.float()).all()
    z2 = eng.sample(c, uc=uc, batch_size=1, shape=(T, 16, H, W), generator=torch.Generator().manual_seed(1), fused=False)
    torch.testing.assert_close(z2.float(), z.float(), rtol=2e-2, atol=2e-2)
    # the oracle's RFSampler + Denoiser + VanillaCFG loop (oracle sample(): sampling.py:950-982) on the same noise and conditions
    x0 = torch.randn(1, T, 16, H, W, generator=torch.Generator().manual_seed(1))
    cpu = lambda t: t.float().cpu()
    want, _ = O.sample(cfg, sd, x0, cpu(c["crossattn"]), cpu(uc["crossattn"]), cpu(shared["ref_concat"]), cpu(shared["concat_smpl_render"]),
                       cpu(shared["image_clip_features"]), num_steps=2, cfg_scale=4.0, shift_scale=5.0)
    # tolerance of the 2-step CFG sampler as in test_sampler_*: the guidance combine v_u + 4 (v_c - v_u) carries (2 * 4 - 1) x the
    # forward error of the bf16 network into every step -> atol 0.14, plus cosine and mean-error bounds
    assert _cos(z.float().cpu(), want) >= 0.999
    assert float((z.float().cpu() - want).abs().mean()) < 2e-2
    torch.testing.assert_close(z.float().cpu(), want, rtol=3e-2, atol=0.14)


@pytest.mark.parametrize("world,mode", [(2, "allgather"), (2, "ulysses")])
def test_sequence_parallel_emulated_equals_single(golden_dir, world, mode):
    """N virtual SP ranks (threads sharing the one GPU, scail_amd.parallel.ThreadBackend) run the real
    multi-rank data path: H-chunked latents, rank-shifted RoPE, K / V^T all-gather, multi-segment
    attention kernel, gather to rank 0.  Must reproduce the reference golden (= SP 1)."""
    import threading
    from scail_amd.parallel import SequenceParallel, ThreadBackend
    g = _load(golden_dir, "dit_tiny.npz")
    shared = ThreadBackend.Shared(world)
    outs, errs = [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            cfg, sd, net = _net(O.TINY, int(g["seed"]))
            sp = SequenceParallel(ThreadBackend(shared, r), mode=mode)
            net.sp = sp
            sp.check_latent(g["x"].shape[3], g["x"].shape[4], 3)
            ch = lambda t: sp.chunk(t.to(DEV), 3)
            o = net(ch(g["x"]), timesteps=g["t"].to(DEV), context=g["ctx"].to(DEV),
                    concat_images=torch.zeros(1, device=DEV), ref_concat=ch(g["ref"]),
                    concat_smpl_render=ch(g["pose"]), image_clip_features=g["clip"].to(DEV), chunk_dim=3)
            outs[r] = sp.gather_to_rank0(o, 3)
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    torch.testing.assert_close(outs[0].float().cpu(), g["out"], rtol=2e-2, atol=2e-2)
    assert _cos(outs[0].float().cpu(), g["out"]) >= 0.999


@pytest.mark.parametrize("mode,chunk_dim", [("ulysses", 3), ("allgather", 3), ("ulysses", 4), ("auto", 3)])
def test_sequence_parallel_four_ranks_vs_single_and_oracle(mode, chunk_dim):
    """Four virtual ranks on a 4-head network (ulysses needs heads % ranks == 0): split along H (chunk_dim 3) and along W
    (chunk_dim 4, RoPE W shift); the gathered result must equal the single-rank run of the same network and the oracle."""
    import threading
    from scail_amd.parallel import SequenceParallel, ThreadBackend
    world = 4
    cfgd = dict(hidden_size=512, num_layers=2, num_attention_heads=4, inner_hidden_size=1024, text_dim=64,
                time_freq_dim=256, time_embed_dim=512, latent_height=64, latent_width=64, num_frames=13)
    gen = torch.Generator().manual_seed(11)
    r = lambda *sh: torch.randn(*sh, generator=gen).to(torch.bfloat16).float()
    T, H, W = (2, 16, 32) if chunk_dim == 3 else (2, 32, 16)
    x, ctx, t = r(2, T, 16, H, W), r(2, 12, 64), torch.tensor([500.0, 500.0])
    ref, pose, clip = r(1, 1, 16, H, W), r(1, T, 16, H // 2, W // 2), r(1, 5, 1280)
    cfg, sd, net1 = _net(cfgd, 77)
    kw = dict(concat_images=torch.zeros(1, device=DEV), image_clip_features=clip.to(DEV))
    single = net1(x.to(DEV), timesteps=t.to(DEV), context=ctx.to(DEV), ref_concat=ref.to(DEV), concat_smpl_render=pose.to(DEV), **kw)
    want = O.dit_forward(cfg, sd, x, t, ctx, ref, pose, clip)
    torch.testing.assert_close(single.float().cpu(), want, rtol=2e-2, atol=2e-2)
    shared = ThreadBackend.Shared(world)
    outs, errs = [None] * world, []

    def run(rk):
        try:
            torch.cuda.set_device(0)
            _, _, net = _net(cfgd, 77)
            sp = SequenceParallel(ThreadBackend(shared, rk), mode=mode)
            net.sp = sp
            assert sp.resolve_mode(4) == ("ulysses" if mode == "auto" else mode)
            sp.check_latent(H, W, chunk_dim)
            ch = lambda tt: sp.chunk(tt.to(DEV), chunk_dim)
            o = net(ch(x), timesteps=t.to(DEV), context=ctx.to(DEV), ref_concat=ch(ref), concat_smpl_render=ch(pose),
                    chunk_dim=chunk_dim, **kw)
            outs[rk] = sp.gather_to_rank0(o, chunk_dim)
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=run, args=(rk,)) for rk in range(world)]
    [tt.start() for tt in th]
    [tt.join() for tt in th]
    assert not errs, errs
    torch.testing.assert_close(outs[0].float().cpu(), single.float().cpu(), rtol=2e-2, atol=2e-2)
    assert _cos(outs[0].float().cpu(), want) >= 0.999


def test_cli_tiny_end_to_end():
    """BASELINE.json configs[0] shape through the CLI driver: VAE encode -> 2-step sampler -> VAE decode."""
    from scail_amd import cli
    video, z, dt = cli.run(cli.TINY, steps=2, frames=13)
    assert z.shape == (1, 16, 4, 8, 8) and video.shape == (1, 3, 13, 64, 64)
    assert torch.isfinite(video).all() and 0.0 <= float(video.min()) and float(video.max()) <= 1.0


def test_request_pipeline_composition_vs_reference_golden(golden_dir, tmp_path, capsys):
    """The encode -> sample -> decode COMPOSITION against the reference's own (e2e_tiny.npz, oracle/gen_golden_e2e.py: the real
    SATVideoDiffusionEngine.encode_first_stage / sample / decode_first_stage with the script lines of sample_video.py:338-391,
    :455-494 between them): reference frame (examples/001/ref.jpg at the tiny size) and a driving clip as FILES through
    cli.request_from_files (loaders, [-1, 1] mapping, half-resolution pose), cli.run (VAE mean x scale factor 0.8, layout
    permutes, 3 CFG steps on the global RNG stream's noise, 1 / scale factor, decode, clamp((x + 1) / 2)).  Values, not shapes."""
    import copy
    from PIL import Image
    from oracle import wan_vae_oracle as V
    from scail_amd import cli
    g = _load(golden_dir, "e2e_tiny.npz")
    H, W = (int(v) for v in g["size"])
    Image.fromarray(g["ref_u8"][0].permute(1, 2, 0).numpy()).save(tmp_path / "ref.png")               # lossless
    np.save(tmp_path / "rendered.npy", g["pose_u8"].numpy())
    torch.save({"context": g["ctx"], "uncond_context": g["uc_ctx"], "clip": g["clip"]}, tmp_path / "cond.pt")
    cfg = copy.deepcopy(cli.TINY)
    cfg["model"]["scale_factor"] = float(g["scale_factor"])
    cfg["model"]["sampler_config"]["params"]["num_steps"] = int(g["steps"])
    cfg["args"]["sampling_image_size"] = [H, W]
    engine = cli.build_engine(cfg)
    dcfg = O.DiTConfig(**O.CONFIG1)
    missing, unexpected = engine.network.load_state_dict(O.make_state_dict(dcfg, seed=int(g["dit_seed"])), strict=True)
    assert not missing and not unexpected
    missing, unexpected = engine.first_stage_model.model.load_state_dict(
        V.make_state_dict(V.VAEConfig(dim=32, z_dim=16), seed=int(g["vae_seed"])), strict=True)
    assert not missing and not unexpected
    req, size = cli.request_from_files(str(tmp_path / "ref.png"), str(tmp_path / "rendered.npy"), cfg, str(tmp_path / "cond.pt"),
                                       text_dim=engine.network.text_dim)
    assert size == (H, W) and req["ref"].shape == (3, 1, H, W) and req["pose"].shape == (3, 13, H // 2, W // 2)
    # stage 1: the two conditioning latents (b c t h w -> b t c h w, scale factor applied)
    ref_concat = engine.encode_first_stage(req["ref"].unsqueeze(0), None, force_encode=True).permute(0, 2, 1, 3, 4)
    smpl = engine.encode_first_stage(req["pose"].unsqueeze(0), None, force_encode=True).permute(0, 2, 1, 3, 4)
    torch.testing.assert_close(ref_concat.float().cpu(), g["ref_concat"], rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(smpl.float().cpu(), g["smpl_render_latent"], rtol=3e-2, atol=3e-2)
    # stage 2 + 3: the driver
    video, z, _ = cli.run(cfg, req, engine=engine, seed=int(g["noise_seed"]))
    assert z.shape == g["samples_z"].shape and video.shape == (1, 3, 13, H, W)
    samples = video.permute(0, 2, 1, 3, 4).float().cpu()                                               # sample_video.py:493
    want_z, want = g["samples_z"], g["samples"].float()
    dz, dv = (z.float().cpu() - want_z).abs(), (samples - want).abs()
    with capsys.disabled():
        print("\npipeline vs reference: latent max %.4f mean %.5f cos %.6f | video max %.4f mean %.5f cos %.6f"
              % (float(dz.max()), float(dz.mean()), _cos(z.float().cpu(), want_z), float(dv.max()), float(dv.mean()), _cos(samples, want)))
    # latents: the sampler tests' bound (CFG carries 7 x the bf16 forward error) on top of conditioning latents that are
    # themselves bf16 VAE outputs
    assert _cos(z.float().cpu(), want_z) >= 0.999
    assert float(dz.mean()) < 2e-2
    torch.testing.assert_close(z.float().cpu(), want_z, rtol=3e-2, atol=0.14)
    # video in [0, 1]: the VAE tests' 3e-2 on [-1, 1] is 1.5e-2 here, plus the latent error through the decoder
    assert _cos(samples, want) >= 0.999
    assert float(dv.mean()) < 1e-2 and float(dv.max()) < 6e-2          # measured (round 6): mean 0.0028, max 0.023; latents mean 0.011
    # a missing scale factor on either side would not pass: 0.8 vs 1.0 on the conditioning latents is a 25 % error
    assert float((ref_concat.float().cpu() / float(g["scale_factor"]) - g["ref_concat"]).abs().mean()) > 0.05


def test_cli_from_files_to_saved_video(tmp_path):
    """The request as files (reference image + driving video) through preprocessing, VAE, sampler, VAE, and the saved result
    read back (sample_video.py:300-351 / :484-507; containers: scail_amd/video_io.py)."""
    import numpy as np
    from PIL import Image
    from scail_amd import cli, video_io
    g = np.random.default_rng(0)
    Image.fromarray(g.integers(0, 255, (80, 120, 3), dtype=np.uint8)).save(tmp_path / "ref.jpg")
    np.save(tmp_path / "rendered.npy", g.integers(0, 255, (13, 70, 90, 3), dtype=np.uint8))
    req = lambda text_dim: cli.request_from_files(str(tmp_path / "ref.jpg"), str(tmp_path / "rendered.npy"), cli.TINY, text_dim=text_dim)[0]
    video, z, _ = cli.run(cli.TINY, req, steps=2)
    assert video.shape == (1, 3, 13, 64, 64) and torch.isfinite(video).all() and 0.0 <= float(video.min()) and float(video.max()) <= 1.0
    samples = video.permute(0, 2, 1, 3, 4).contiguous().cpu()
    p = video_io.save_multi_video_grid([samples], str(tmp_path / "out"), fps=16, key="0_output")[0]
    back = video_io.load_video_for_pose_sample(p)
    assert back.shape == (13, 64, 64, 3)
    assert np.array_equal(back.numpy(), (255.0 * samples[0].permute(0, 2, 3, 1)).numpy().astype(np.uint8))


def test_cli_request_line_command(tmp_path):
    """`python -m scail_amd.cli --tiny --request '<prompt>@@<example_dir>'`: the reference's request format end to end, results in
    <output-dir>/<example>/ (sample_video.py:284-300, :410-414, :506)."""
    import subprocess
    import sys
    import numpy as np
    from PIL import Image
    from scail_amd import video_io
    d = tmp_path / "001"
    d.mkdir()
    g = np.random.default_rng(1)
    Image.fromarray(g.integers(0, 255, (72, 96, 3), dtype=np.uint8)).save(d / "ref.png")
    np.save(d / "rendered.npy", g.integers(0, 255, (9, 72, 96, 3), dtype=np.uint8))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "scail_amd.cli", "--tiny", "--steps", "2", "--request", f"a girl is dancing@@{d}",
                        "--output-dir", str(tmp_path / "out"), "--format", ".png"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "out" / "001" / "text.txt").read_text() == "a girl is dancing"
    back = video_io.load_video_for_pose_sample(str(tmp_path / "out" / "001" / "001_output_000000.png"))
    assert back.shape == (9, 64, 64, 3)


def test_cli_prompt_to_conditioning(tmp_path):
    """Prompt string + reference image -> UMT5 / CLIP conditioning -> sampler, all on the GPU (sample_video.py:397-438).
    Tiny text width; SentencePiece model trained here; CLIP tower at full size with random weights."""
    import io
    spm = pytest.importorskip("sentencepiece")
    from scail_amd import cli
    buf = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(["the girl is dancing in the street", "a man walks his dog"] * 40), model_writer=buf,
                                   vocab_size=40, model_type="unigram", hard_vocab_limit=False, minloglevel=2, pad_id=0, eos_id=1, unk_id=2, bos_id=-1)
    (tmp_path / "spiece.model").write_bytes(buf.getvalue())
    import copy
    cfg = copy.deepcopy(cli.TINY)
    cfg["model"]["network_config"]["params"]["text_dim"] = 128
    req = cli.synthetic_request(64, 64, 13, 128, 12, DEV, 0)
    cond = cli.encode_conditioning("the girl is dancing", "", req["ref"], 128, str(tmp_path / "spiece.model"), max_length=16)
    assert cond["context"].shape == (1, 16, 128) and cond["uncond_context"].shape == (1, 16, 128) and cond["clip"].shape == (1, 257, 1280)
    assert float(cond["uncond_context"].float()[0, 1:].abs().max()) == 0.0          # "" is one </s> row, the rest zeroed
    assert float(cond["context"].float()[0, :3].abs().min(-1).values.max()) >= 0.0 and torch.isfinite(cond["clip"].float()).all()
    req.update(cond)
    video, z, _ = cli.run(cfg, req, steps=2)
    assert video.shape == (1, 3, 13, 64, 64) and torch.isfinite(video).all()
    with pytest.raises(ValueError, match="multiple of 128"):
        cli.encode_conditioning("x", "", req["ref"], 64, str(tmp_path / "spiece.model"))


def test_engine_conditioner_from_reference_style_config(tmp_path):
    """``conditioner_config`` / ``i2v_clip_config`` with the reference's target strings build the MI355X encoders behind the
    reference's GeneralConditioner protocol: get_batch -> get_unconditional_conditioning -> engine.sample
    (sample_video.py:397-400, :416-438, :476-483)."""
    import copy
    import io
    spm = pytest.importorskip("sentencepiece")
    from scail_amd import cli
    from scail_amd.conditioner import get_batch, get_unique_embedder_keys_from_conditioner
    from scail_amd.engine import SATVideoDiffusionEngine
    buf = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(["the girl is dancing in the street", "a man walks his dog"] * 40), model_writer=buf,
                                   vocab_size=40, model_type="unigram", hard_vocab_limit=False, minloglevel=2, pad_id=0, eos_id=1, unk_id=2, bos_id=-1)
    (tmp_path / "spiece.model").write_bytes(buf.getvalue())
    mc = copy.deepcopy(cli.TINY["model"])
    mc["network_config"]["params"]["text_dim"] = 128
    mc.update(build_conditioner=True, build_i2v_clip=True, build_first_stage=False,
              conditioner_config={"target": "sgm.modules.GeneralConditioner", "params": {"emb_models": [
                  {"is_trainable": False, "input_key": "txt", "ucg_rate": 0.1, "legacy_ucg_val": "",
                   "target": "sgm.modules.encoders.umt5.T5EncoderModel",
                   "params": dict(tokenizer_path=str(tmp_path / "spiece.model"), max_length=16, vocab=64, dim=128, dim_attn=128,
                                  dim_ffn=256, num_heads=2, num_layers=2)}]}},
              i2v_clip_config={"target": "sgm.modules.encoders.clip.CLIPModel", "params": dict(num_layers=2)})
    eng = SATVideoDiffusionEngine(mc, device=DEV)
    assert eng.conditioner is not None and eng.i2v_clip is not None and eng.use_i2v_clip
    batch, batch_uc = get_batch(get_unique_embedder_keys_from_conditioner(eng.conditioner),
                                {"prompt": "the girl is dancing", "negative_prompt": "", "num_frames": torch.tensor([4])}, [1])
    c, uc = eng.conditioner.get_unconditional_conditioning(batch, batch_uc=batch_uc, force_uc_zero_embeddings=[])
    assert set(c) == {"crossattn"} and c["crossattn"].shape == (1, 16, 128) and uc["crossattn"].shape == (1, 16, 128)
    assert float((c["crossattn"].float() - uc["crossattn"].float()).abs().max()) > 0
    _, uc0 = eng.conditioner.get_unconditional_conditioning(batch, batch_uc=batch_uc, force_uc_zero_embeddings=["txt"])
    assert float(uc0["crossattn"].float().abs().max()) == 0.0
    g = torch.Generator().manual_seed(3)
    ref_img = (torch.rand(1, 3, 1, 64, 64, generator=g) * 2 - 1).to(DEV)                     # b c t h w
    feats = eng.i2v_clip.visual(ref_img)
    assert feats.shape == (1, 257, 1280)
    shared = dict(concat_images=torch.zeros(1, device=DEV), image_clip_features=feats.to(torch.bfloat16),
                  ref_concat=torch.randn(1, 1, 16, 8, 8, generator=g).to(DEV).to(torch.bfloat16),
                  concat_smpl_render=torch.randn(1, 4, 16, 4, 4, generator=g).to(DEV).to(torch.bfloat16))
    z = eng.sample(dict(c, **shared), uc=dict(uc, **shared), batch_size=1, shape=(4, 16, 8, 8), num_steps=2)
    assert z.shape == (1, 4, 16, 8, 8) and torch.isfinite(z.float()).all()


def test_multi_character_extension_vs_oracle(golden_dir, n_char=2):
    """BASELINE config 5 (multi-character in-context concat) is NOT in the reference (one reference frame, one pose stream,
    dit...:1559): an extension with token order [ref_0..ref_{C-1} | noise | pose_0..pose_{C-1}] and the RoPE windows of
    rope.build_tables(n_char=C).  Checked against the oracle extended the same way (parity unpinned by construction);
    C = 1 through the same code equals the reference golden; the extra character changes the result."""
    g = _load(golden_dir, "dit_tiny.npz")
    cfg, sd, net = _net(O.TINY, int(g["seed"]))
    T = g["x"].shape[1]
    gen = torch.Generator().manual_seed(77)
    refs = torch.cat([g["ref"]] + [torch.randn(g["ref"].shape, generator=gen) for _ in range(n_char - 1)], 1)
    poses = torch.cat([g["pose"]] + [torch.randn(g["pose"].shape, generator=gen) for _ in range(n_char - 1)], 1)
    assert refs.shape[1] == n_char and poses.shape[1] == n_char * T
    want, oh = O.dit_forward(cfg, sd, g["x"], g["t"], g["ctx"], refs, poses, g["clip"], return_hidden=True)
    hidden = {}
    net._tap = lambda i, h: hidden.__setitem__(i, h.float().cpu().clone())
    kw = dict(concat_images=torch.zeros(1, *g["x"].shape[1:], device=DEV), image_clip_features=g["clip"].to(DEV))
    out = net.forward_f32(g["x"].to(DEV), g["t"].to(DEV), g["ctx"].to(DEV), None, ref_concat=refs.to(DEV),
                          concat_smpl_render=poses.to(DEV), **kw)
    assert out.shape == g["out"].shape
    for i in range(-1, cfg.num_layers):
        torch.testing.assert_close(hidden[i], oh[i + 1], rtol=2e-2, atol=2e-2, msg=lambda m: f"block {i}: {m}")
    torch.testing.assert_close(out.cpu(), want, rtol=2e-2, atol=2e-2)
    assert _cos(out.cpu(), want) >= 0.999
    net._tap = None
    one = net.forward_f32(g["x"].to(DEV), g["t"].to(DEV), g["ctx"].to(DEV), None, ref_concat=g["ref"].to(DEV),
                          concat_smpl_render=g["pose"].to(DEV), **kw)
    torch.testing.assert_close(one.cpu(), g["out"], rtol=2e-2, atol=2e-2)
    assert (one - out).abs().max() > 1e-2
    with pytest.raises(Exception, match="frames"):
        net.forward_f32(g["x"].to(DEV), g["t"].to(DEV), g["ctx"].to(DEV), None, ref_concat=refs.to(DEV),
                        concat_smpl_render=g["pose"].to(DEV), **kw)
    # a third character's reference window (2 * global_rope_W) lies outside the reference's RoPE table extent
    with pytest.raises(ValueError, match="table extent"):
        net.forward_f32(g["x"].to(DEV), g["t"].to(DEV), g["ctx"].to(DEV), None, ref_concat=torch.cat([refs, refs[:, :1]], 1).to(DEV),
                        concat_smpl_render=torch.cat([poses, poses[:, :T]], 1).to(DEV), **kw)

"""Generate tests/golden/*.npz by running the REAL reference on CPU (fp32).

Build-container only (needs /root/reference).  Usage:  python oracle/gen_golden.py

The reference ships no tests or golden vectors, so the fixtures the
parity tests use are outputs of the reference's own classes on seeded inputs:

  dit_tiny.npz      DiffusionTransformer.forward (dit...:1452-1587) incl. per-layer hidden states,
                    2-layer / 256-dim / 2-head config, weights = oracle.make_state_dict(seed)
  dit_config1.npz   the same for BASELINE.json configs[0] (2-layer / 128-dim / 1 head)
  dit_tiny_sp.npz   the same network evaluated the way sequence-parallel rank r of 2 sees it
                    for the *embedding/rope* part (rope_H_shift, dit...:1578-1585)
  rope_tiny.npz     Rotary3DPositionEmbeddingMixin.rotary/_ref/_pose on a random tensor
  sampler_tiny.npz  RFSampler + Denoiser(RFScaling) + VanillaCFG + OpenAIWrapper, 2 steps
  sigmas50.npz      50-step schedule (sampling.py:888-903)
  sampler_tiny_50.npz  the SHIPPED step count (yaml :113-131: 50 steps, shift 5, CFG 4) on BASELINE config 1's network
                    (2-layer / 128-dim, 4x8x8 latent): final latent + the latents after steps 2 / 10 / 25 (the drift curve
                    of a bf16 network is measured against these)
  sampler_long_tiny.npz  RFSamplerLong (sampling.py:986-1085): 6-frame latent, three overlapping 4-frame tiles, 2 steps
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_shims  # noqa: E402
from oracle import scail_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def tiny_inputs(seed=7, B=2, T=4, H=8, W=8, Lt=12, n_clip=5, text_dim=64, n_cond=1):
    g = torch.Generator().manual_seed(seed)
    x = bf16r(torch.randn(B, T, 16, H, W, generator=g))
    ctx = bf16r(torch.randn(B, Lt, text_dim, generator=g))
    ctx[:, Lt // 2:] = 0            # zeroed padding rows, umt5.py:516-522
    ref = bf16r(torch.randn(n_cond, 1, 16, H, W, generator=g))
    pose = bf16r(torch.randn(n_cond, T, 16, H // 2, W // 2, generator=g))
    clip = bf16r(torch.randn(n_cond, n_clip, 1280, generator=g))
    t = torch.tensor([731.0, 731.0])
    return dict(x=x, ctx=ctx, ref=ref, pose=pose, clip=clip, t=t)


def gen_dit_tiny(name="dit_tiny", cfgd=None, inputs=None):
    cfg = O.DiTConfig(**(cfgd or O.TINY))
    sd = O.make_state_dict(cfg, seed=1234)
    net = ref_shims.build_reference_dit(cfg, sd)
    inp = inputs or tiny_inputs()
    hidden = []
    mix = net.mixins["adaln_layer"]
    orig = mix.layer_forward

    def tap(*a, **k):           # SAT calls the mixin hook directly (transformer.py:712-719)
        o = orig(*a, **k)
        hidden.append(o.detach().clone())
        return o

    mix.layer_forward = tap
    net.collect_hooks_()        # hooks are snapshotted at add_mixin time (base_model.py:140-176)
    with torch.no_grad():
        out = net(inp["x"], timesteps=inp["t"], context=inp["ctx"],
                  concat_images=torch.zeros(1, *inp["x"].shape[1:]), ref_concat=inp["ref"],
                  concat_smpl_render=inp["pose"], image_clip_features=inp["clip"])
    del mix.layer_forward
    net.collect_hooks_()
    assert len(hidden) == cfg.num_layers
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=1234,
                        **{k: v.numpy() for k, v in inp.items()}, out=out.numpy(),
                        **{f"hidden{i + 1}": h.numpy() for i, h in enumerate(hidden)})
    print(name + ": out", tuple(out.shape), "abs-mean", float(out.abs().mean()))

    if name != "dit_tiny":
        return cfg, sd, net, inp
    # sequence-parallel view: rank r of 2 holds rows [r*H/2, (r+1)*H/2) of every latent and
    # shifts its RoPE window by r*(H/2/patch) (diffusion_video.py:495-503, dit...:1578-1585).
    # The reference mixin is driven directly (no process group of size 2 needed) to pin the
    # shifted rope tables + patch embedding for a shard.
    ref = ref_shims.load_reference()
    pos = net.mixins["pos_embed"]
    g = torch.Generator().manual_seed(11)
    H = inp["x"].shape[3]
    res = {}
    for r in range(2):
        hs = H // 2
        kw = dict(rope_T=4, rope_H=hs // 2, rope_W=4, rope_H_shift=r * (hs // 2), rope_W_shift=0,
                  global_rope_H=0, global_rope_W=120)
        Lr, Ln, Lp = kw["rope_H"] * 4, 4 * kw["rope_H"] * 4, 4 * (kw["rope_H"] // 2) * 2
        q = torch.randn(1, 2, Lr + Ln + Lp, cfg.head_dim, generator=g)
        with torch.no_grad():
            qr = torch.cat([pos.rotary_ref(q[:, :, :Lr], **kw), pos.rotary(q[:, :, Lr:Lr + Ln], **kw),
                            pos.rotary_pose(q[:, :, -Lp:], **kw)], dim=2)
        res[f"q{r}"] = q.numpy()
        res[f"qr{r}"] = qr.numpy()
    np.savez_compressed(os.path.join(OUT, "rope_tiny_sp.npz"), **res)
    return cfg, sd, net, inp


def gen_rope(cfg, net):
    pos = net.mixins["pos_embed"]
    g = torch.Generator().manual_seed(3)
    kw = dict(rope_T=3, rope_H=6, rope_W=4, rope_H_shift=0, rope_W_shift=0, global_rope_H=0, global_rope_W=120)
    Lr, Ln, Lp = 6 * 4, 3 * 6 * 4, 3 * 3 * 2
    q = torch.randn(2, 2, Lr + Ln + Lp, cfg.head_dim, generator=g)
    with torch.no_grad():
        qr = torch.cat([pos.rotary_ref(q[:, :, :Lr], **kw), pos.rotary(q[:, :, Lr:Lr + Ln], **kw),
                        pos.rotary_pose(q[:, :, -Lp:], **kw)], dim=2)
    np.savez_compressed(os.path.join(OUT, "rope_tiny.npz"), q=q.numpy(), qr=qr.numpy(),
                        rope_T=3, rope_H=6, rope_W=4)
    print("rope_tiny:", tuple(qr.shape))


def gen_sampler(cfg, sd, net, inp):
    ref = ref_shims.load_reference()
    sampling, denoiser_mod, wrappers, guiders = ref["sampling"], ref["denoiser"], ref["wrappers"], ref["guiders"]
    from sgm.modules.diffusionmodules.denoiser_scaling import RFScaling
    from sgm.modules.diffusionmodules.denoiser_weighting import EpsWeighting

    class _Den(denoiser_mod.Denoiser):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.weighting = EpsWeighting()
            self.scaling = RFScaling()

    den = _Den()
    wrapped = wrappers.OpenAIWrapper(net, compile_model=False, dtype=torch.float32)
    sampler = sampling.RFSampler(
        schedule_shift=False, hunyuan_schedule=True, shift_scale=5, mode="normal", num_steps=2, verbose=False,
        device="cpu",
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.RFDiscretization",
                               "params": {"reverse": False}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})
    g = torch.Generator().manual_seed(21)
    x0 = torch.randn(1, *inp["x"].shape[1:], generator=g)
    uc_ctx = torch.zeros_like(inp["ctx"][:1])
    uc_ctx[:, :1] = bf16r(torch.randn(1, 1, inp["ctx"].shape[-1], generator=g))
    c_ctx = inp["ctx"][1:2]
    shared = dict(concat_images=torch.zeros(1, *inp["x"].shape[1:]), ref_concat=inp["ref"],
                  concat_smpl_render=inp["pose"], image_clip_features=inp["clip"])
    c = dict(crossattn=c_ctx.clone(), **{k: v.clone() for k, v in shared.items()})
    uc = dict(crossattn=uc_ctx.clone(), **{k: v.clone() for k, v in shared.items()})
    # diffusion_video.py:555-563
    fn = lambda inp_, sigma, cc, **kw: den(wrapped, inp_, sigma, cc, concat_images=None, chunk_dim=None, **kw)
    with torch.no_grad():
        xT = sampler(fn, x0.clone(), c, uc=uc)
    sig = sampling.make_flow_timesteps(0, 2, verbose=False, shift_scale=5, mode="normal")
    np.savez_compressed(os.path.join(OUT, "sampler_tiny.npz"), x0=x0.numpy(), uc_ctx=uc_ctx.numpy(),
                        c_ctx=c_ctx.numpy(), xT=xT.numpy(), sigmas=sig.numpy())
    sig50 = sampling.make_flow_timesteps(0, 50, verbose=False, shift_scale=5, mode="normal")
    np.savez_compressed(os.path.join(OUT, "sigmas50.npz"), sigmas=sig50.numpy())
    print("sampler_tiny: xT abs-mean", float(xT.abs().mean()), "sigmas", sig.tolist())


def gen_sampler50():
    """The step count config 2 actually runs (configs/video_model/Wan2.1-i2v-14Bsc-pose-xc-latent.yaml:113-131:
    num_steps 50, shift_scale 5, VanillaCFG scale 4) through the real RFSampler.__call__ (sampling.py:965-982) +
    Denoiser + OpenAIWrapper + the config-1 DiT, fp32 on CPU.  The latent after k steps is what the denoiser
    receives at call k (sampler_step :960-963 is the only writer), recorded for k = 2, 10, 25."""
    ref = ref_shims.load_reference()
    sampling, denoiser_mod, wrappers = ref["sampling"], ref["denoiser"], ref["wrappers"]
    from sgm.modules.diffusionmodules.denoiser_scaling import RFScaling
    from sgm.modules.diffusionmodules.denoiser_weighting import EpsWeighting
    cfg = O.DiTConfig(**O.CONFIG1)
    sd = O.make_state_dict(cfg, seed=1234)
    net = ref_shims.build_reference_dit(cfg, sd)
    inp = tiny_inputs()

    class _Den(denoiser_mod.Denoiser):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.weighting = EpsWeighting()
            self.scaling = RFScaling()

    den = _Den()
    wrapped = wrappers.OpenAIWrapper(net, compile_model=False, dtype=torch.float32)
    sampler = sampling.RFSampler(
        schedule_shift=False, hunyuan_schedule=True, shift_scale=5, mode="normal", num_steps=50, verbose=False,
        device="cpu",
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.RFDiscretization",
                               "params": {"reverse": False}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})
    g = torch.Generator().manual_seed(50)
    x0 = torch.randn(1, *inp["x"].shape[1:], generator=g)
    uc_ctx = torch.zeros_like(inp["ctx"][:1])
    uc_ctx[:, :1] = bf16r(torch.randn(1, 1, inp["ctx"].shape[-1], generator=g))
    c_ctx = inp["ctx"][1:2]
    shared = dict(concat_images=torch.zeros(1, *inp["x"].shape[1:]), ref_concat=inp["ref"],
                  concat_smpl_render=inp["pose"], image_clip_features=inp["clip"])
    c = dict(crossattn=c_ctx.clone(), **{k: v.clone() for k, v in shared.items()})
    uc = dict(crossattn=uc_ctx.clone(), **{k: v.clone() for k, v in shared.items()})
    seen = []

    def fn(inp_, sigma, cc, **kw):          # diffusion_video.py:555-563
        seen.append(inp_[:1].detach().clone())          # CFG batch = [x; x] (guiders.py:47-57)
        return den(wrapped, inp_, sigma, cc, concat_images=None, chunk_dim=None, **kw)

    with torch.no_grad():
        xT = sampler(fn, x0.clone(), c, uc=uc)
    assert len(seen) == 50 and torch.equal(seen[0], x0)
    np.savez_compressed(os.path.join(OUT, "sampler_tiny_50.npz"), seed=1234, x0=x0.numpy(), uc_ctx=uc_ctx.numpy(),
                        c_ctx=c_ctx.numpy(), ref=inp["ref"].numpy(), pose=inp["pose"].numpy(), clip=inp["clip"].numpy(),
                        x2=seen[2].numpy(), x10=seen[10].numpy(), x25=seen[25].numpy(), xT=xT.numpy())
    print("sampler_tiny_50: xT abs-mean", float(xT.abs().mean()), "x0 abs-mean", float(x0.abs().mean()))


def gen_sampler_long(cfg, sd, net, inp):
    """RFSamplerLong (sampling.py:986-1085) on a 6-frame latent, two overlapping 4-frame tiles, 2 steps."""
    ref = ref_shims.load_reference()
    sampling, denoiser_mod, wrappers = ref["sampling"], ref["denoiser"], ref["wrappers"]
    from sgm.modules.diffusionmodules.denoiser_scaling import RFScaling
    from sgm.modules.diffusionmodules.denoiser_weighting import EpsWeighting

    class _Den(denoiser_mod.Denoiser):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.weighting = EpsWeighting()
            self.scaling = RFScaling()

    den = _Den()
    wrapped = wrappers.OpenAIWrapper(net, compile_model=False, dtype=torch.float32)
    sampler = sampling.RFSamplerLong(
        schedule_shift=False, hunyuan_schedule=True, shift_scale=5, mode="normal", num_steps=2, verbose=False,
        device="cpu",
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.RFDiscretization",
                               "params": {"reverse": False}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})
    g = torch.Generator().manual_seed(33)
    T, Tt = 6, 4
    tiles = [[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 5]]
    H, W = inp["x"].shape[-2:]
    x0 = torch.randn(1, T, 16, H, W, generator=g)
    smpl_tiled = bf16r(torch.randn(1, len(tiles), Tt, 16, H // 2, W // 2, generator=g))
    uc_ctx = torch.zeros_like(inp["ctx"][:1])
    uc_ctx[:, :1] = bf16r(torch.randn(1, 1, inp["ctx"].shape[-1], generator=g))
    c_ctx = inp["ctx"][1:2]
    shared = dict(concat_images=torch.zeros(1, Tt, 16, H, W), ref_concat=inp["ref"], smpl_tiled=smpl_tiled,
                  image_clip_features=inp["clip"])
    c = dict(crossattn=c_ctx.clone(), **{k: v.clone() for k, v in shared.items()})
    uc = dict(crossattn=uc_ctx.clone(), **{k: v.clone() for k, v in shared.items()})
    fn = lambda inp_, sigma, cc, **kw: den(wrapped, inp_, sigma, cc, concat_images=None, chunk_dim=None, **kw)
    with torch.no_grad():
        xT = sampler(fn, x0.clone(), c, uc=uc, tile_indices=tiles)
    np.savez_compressed(os.path.join(OUT, "sampler_long_tiny.npz"), x0=x0.numpy(), uc_ctx=uc_ctx.numpy(),
                        c_ctx=c_ctx.numpy(), smpl_tiled=smpl_tiled.numpy(), tiles=np.array(tiles), xT=xT.numpy())
    print("sampler_long_tiny: xT abs-mean", float(xT.abs().mean()))


def gen_rope_sp_w():
    """Sequence-parallel W split (portrait latents: chunk_dim 4, diffusion_video.py:504-552): rank r of 2 holds columns
    [r*W/2, (r+1)*W/2) and shifts its RoPE window by r*(W/2/patch) (dit...:1583-1585).  Drives the reference mixin directly like the
    H-split case in gen_dit_tiny.  8 x 12 patches split into two 8 x 6 slabs, 2 frames."""
    cfg = O.DiTConfig(**O.TINY)
    net = ref_shims.build_reference_dit(cfg, O.make_state_dict(cfg, seed=1234))
    pos = net.mixins["pos_embed"]
    g = torch.Generator().manual_seed(12)
    res = {}
    for r in range(2):
        kw = dict(rope_T=2, rope_H=8, rope_W=6, rope_H_shift=0, rope_W_shift=r * 6, global_rope_H=0, global_rope_W=120)
        Lr, Ln, Lp = 8 * 6, 2 * 8 * 6, 2 * 4 * 3
        q = torch.randn(1, 1, Lr + Ln + Lp, cfg.head_dim, generator=g)
        with torch.no_grad():
            qr = torch.cat([pos.rotary_ref(q[:, :, :Lr], **kw), pos.rotary(q[:, :, Lr:Lr + Ln], **kw),
                            pos.rotary_pose(q[:, :, -Lp:], **kw)], dim=2)
        res[f"q{r}"] = q.numpy()
        res[f"qr{r}"] = qr.numpy()
    np.savez_compressed(os.path.join(OUT, "rope_tiny_sp_w.npz"), **res)
    print("rope_tiny_sp_w written")


def gen_dit_shapes():
    """Ragged everything: portrait latent (12 x 8 -> 6 x 4 patches, pose 3 x 2), odd frame count, 77 text tokens (two key tiles,
    the second ragged), 129 = 2 x 64 + 1 CLIP tokens (a one-key ragged tile like the real 257), conditioning tensors given per batch element (the n == B branch
    of the CFG repeat, dit...:1479-1495)."""
    return gen_dit_tiny("dit_shapes", O.TINY, tiny_inputs(seed=21, B=2, T=3, H=12, W=8, Lt=77, n_clip=129, n_cond=2))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    gen_dit_shapes()
    gen_rope_sp_w()
    gen_dit_tiny("dit_config1", O.CONFIG1)
    cfg, sd, net, inp = gen_dit_tiny()
    gen_rope(cfg, net)
    gen_sampler(cfg, sd, net, inp)
    gen_sampler_long(cfg, sd, net, inp)
    gen_sampler50()


if __name__ == "__main__":
    main()

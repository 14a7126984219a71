"""Generate tests/golden/e2e_tiny.npz: the COMPOSITION of the request pipeline, run with the reference's own pieces the
way sample_video.py composes them, on CPU in fp32.  Build-container only.  Usage: python oracle/gen_golden_e2e.py

What is the reference's own code here (called, not restated):
  * ``SATVideoDiffusionEngine.encode_first_stage`` / ``decode_first_stage`` / ``sample`` (diffusion_video.py:298-331, :456-587),
    unbound methods on an engine object assembled without its constructor (the constructor needs SAT checkpoints / DeepSpeed
    arguments); its members are the reference's ``WanVAE`` wrapper (sgm/models/wan_vae.py:619-668) around a seeded ``WanVAE_``
    (dim 32), the reference ``DiffusionTransformer`` (BASELINE config 1: 2 layers / 128 wide) in ``OpenAIWrapper``, ``Denoiser``
    (RFScaling) and ``RFSampler`` (3 steps, shift 5, VanillaCFG 4);
  * the lines of ``sample_video.py`` between them are inline script code, repeated here statement by statement with the
    line they stand for: pixel normalisation (:340-341), the 0.5x bilinear pose downsample (:350-351), the bf16 rounding of
    the pixels (:356-359; the CPU run continues in fp32), b t c h w <-> b c t h w permutes (:363-367, :381-382), c / uc
    assembly (:455-470), ``sample_func`` (:476-483), permute, decode, permute, clamp((x + 1) / 2) (:484-494).

What is NOT the reference here: ``resize_for_rectangle_crop`` (data_video.py:141-170) needs torchvision, which this image
lacks; the reference frame (examples/001/ref.jpg, through PIL: centre crop to the target aspect, LANCZOS downscale) and the
synthetic driving clip are produced AT the target size, where that function is the identity, and stored in the fixture as
the uint8 / [-1, 1] tensors sample_video.py holds after its crop.  The scale factor is 0.8 (the shipped yamls say 1.0,
which would hide a missing multiply or divide)."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_shims  # noqa: E402
from oracle import scail_oracle as O  # noqa: E402
from oracle import wan_vae_oracle as V  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SIZE = (64, 96)            # sampling_image_size (H, W) of a landscape request -> latents 8 x 12, pose latents 4 x 6
FRAMES = 13                # -> 4 latent frames
SCALE_FACTOR = 0.8
DIT_SEED, VAE_SEED, NOISE_SEED = 1234, 4321, 2024


def reference_frame(size_hw):
    """examples/001/ref.jpg -> uint8 (1, 3, H, W) at the target size (centre crop to the aspect, LANCZOS)."""
    from PIL import Image
    im = Image.open(os.path.join(ref_shims.REFERENCE_ROOT, "examples", "001", "ref.jpg")).convert("RGB")
    th, tw = size_hw
    w, h = im.size
    if w / h > tw / th:
        cw = int(round(h * tw / th))
        im = im.crop(((w - cw) // 2, 0, (w - cw) // 2 + cw, h))
    else:
        ch = int(round(w * th / tw))
        im = im.crop((0, (h - ch) // 2, w, (h - ch) // 2 + ch))
    im = im.resize((tw, th), Image.LANCZOS)
    return torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).unsqueeze(0).contiguous()


def driving_clip(frames, size_hw, seed=9):
    """A synthetic rendered-pose clip, uint8 (T, H, W, 3) like ``load_video_for_pose_sample`` returns: a few coloured
    blobs moving over a dark background."""
    H, W = size_hw
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    out = np.zeros((frames, H, W, 3), np.float32)
    for _ in range(5):
        cy, cx, vy, vx = g.uniform(0.2, 0.8) * H, g.uniform(0.2, 0.8) * W, g.uniform(-1.5, 1.5), g.uniform(-2, 2)
        rad, col = g.uniform(4, 10), g.uniform(60, 255, 3)
        for t in range(frames):
            d2 = (yy - cy - vy * t) ** 2 + (xx - cx - vx * t) ** 2
            out[t] += np.exp(-d2 / (2 * rad * rad))[..., None] * col
    return torch.from_numpy(np.clip(out, 0, 255).astype(np.uint8))


def build_reference_engine():
    ref = ref_shims.load_reference()
    import diffusion_video
    from sgm.modules.diffusionmodules.denoiser_scaling import RFScaling
    from sgm.modules.diffusionmodules.denoiser_weighting import EpsWeighting
    sampling, denoiser_mod, wrappers, wan_vae = ref["sampling"], ref["denoiser"], ref["wrappers"], ref["wan_vae"]
    cfg = O.DiTConfig(**O.CONFIG1)
    net = ref_shims.build_reference_dit(cfg, O.make_state_dict(cfg, seed=DIT_SEED))       # also initialises sat.mpu (1 rank)

    vcfg = V.VAEConfig(dim=32, z_dim=16)
    inner = wan_vae.WanVAE_(dim=vcfg.dim, z_dim=vcfg.z_dim, dim_mult=list(vcfg.dim_mult), num_res_blocks=2, attn_scales=[],
                            temperal_downsample=list(vcfg.temperal_downsample), dropout=0.0).eval()
    missing, unexpected = inner.load_state_dict(V.make_state_dict(vcfg, seed=VAE_SEED), strict=True)
    assert not missing and not unexpected
    vae = wan_vae.WanVAE.__new__(wan_vae.WanVAE)          # the wrapper's constructor loads a checkpoint file (:641-645)
    vae.dtype, vae.device = torch.float32, "cpu"
    vae.mean, vae.std = torch.tensor(V.LATENT_MEAN), torch.tensor(V.LATENT_STD)
    vae.scale = [vae.mean, 1.0 / vae.std]                 # :638-640
    vae.model = inner

    class _Den(denoiser_mod.Denoiser):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.weighting = EpsWeighting()
            self.scaling = RFScaling()

    Engine = diffusion_video.SATVideoDiffusionEngine
    eng = Engine.__new__(Engine)
    torch.nn.Module.__init__(eng)
    eng.first_stage_model = vae
    eng.scale_factor, eng.latent_input, eng.en_and_decode_n_samples_a_time = SCALE_FACTOR, True, None
    eng.model = wrappers.OpenAIWrapper(net, compile_model=False, dtype=torch.float32)
    eng.denoiser = _Den()
    eng.sampler = sampling.RFSampler(
        schedule_shift=False, hunyuan_schedule=True, shift_scale=5, mode="normal", num_steps=3, verbose=False, device="cpu",
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.RFDiscretization", "params": {"reverse": False}},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})
    eng.loss_fn = types.SimpleNamespace()
    eng.dtype, eng.device = torch.float32, "cpu"
    eng.i2v_encode_video, eng.use_i2v_clip = True, True
    return eng


def main():
    from einops import rearrange
    model = build_reference_engine()
    target_H, target_W = SIZE
    ref_u8 = reference_frame(SIZE)                                                   # (1, 3, H, W) uint8
    pose_u8 = driving_clip(FRAMES, SIZE)                                             # (T, H, W, 3) uint8
    g = torch.Generator().manual_seed(77)
    bf = lambda t: t.to(torch.bfloat16).float()
    ctx = bf(torch.randn(1, 16, 64, generator=g))
    ctx[:, 7:] = 0                                                                   # padded rows zeroed (umt5.py:516-522)
    uc_ctx = torch.zeros(1, 16, 64)
    uc_ctx[:, :1] = bf(torch.randn(1, 1, 64, generator=g))
    clip = bf(torch.randn(1, 9, 1280, generator=g))

    # ---- sample_video.py, statement by statement ----
    image_tensor = (ref_u8.float() / 255.0) * 2 - 1                                   # load_image_to_tensor_chw_normalized (:35-45: ToTensor, * 2 - 1)
    pose_video = pose_u8.permute(0, 3, 1, 2).float()                                  # :339  T H W C -> T C H W
    # :340 / :343 resize_for_rectangle_crop: identity at the target size (see the header)
    pose_video = (pose_video - 127.5) / 127.5                                         # :341
    smpl_render_video = pose_video                                                    # :347
    smpl_render_video = F.interpolate(smpl_render_video, scale_factor=0.5, mode="bilinear", align_corners=False)    # :350-351
    smpl_render_video = bf(smpl_render_video.unsqueeze(0))                            # :358  B T C H W, bf16 pixels
    ori_image = bf(image_tensor.unsqueeze(0))                                         # :359  B 1 C H W
    with torch.no_grad():
        ref_concat = model.encode_first_stage(rearrange(ori_image, "b t c h w -> b c t h w").contiguous(), None, force_encode=True)   # :366
        ref_concat = ref_concat.permute(0, 2, 1, 3, 4).contiguous()                   # :367
        smpl_render_latent = model.encode_first_stage(rearrange(smpl_render_video, "b t c h w -> b c t h w").contiguous(), None,
                                                      force_encode=True)            # :381
        smpl_render_latent = smpl_render_latent.permute(0, 2, 1, 3, 4).contiguous()   # :382
        pose_latent = smpl_render_latent                                              # :383
        T = pose_latent.shape[1]                                                      # :388
        C, H, W = ref_concat.shape[2], ref_concat.shape[3], ref_concat.shape[4]       # :389 (image == ref frame + zeros: same C, H, W)
        image = torch.zeros(1, T, C, H, W)                                            # :362-365: only its presence is read (dit...:1457)
        c, uc = {"crossattn": ctx.clone()}, {"crossattn": uc_ctx.clone()}             # :433-447 (conditioner output)
        for d in (c, uc):                                                             # :455-470
            d["concat_images"] = image
            d["ref_concat"] = ref_concat
            d["concat_pose"] = pose_latent
            d["concat_smpl_render"] = smpl_render_latent
            d["image_clip_features"] = clip
        torch.manual_seed(NOISE_SEED)                                                 # the noise is torch.randn on the global stream (:470)
        samples_z = model.sample(c, uc=uc, batch_size=1, shape=(T, C, H, W), ofs=torch.tensor([2.0]), fps=torch.tensor([16]))   # :476-483
        samples_z = samples_z.permute(0, 2, 1, 3, 4).contiguous()                     # :485
        samples_x = model.decode_first_stage(samples_z).to(torch.float32)             # :491
        samples_x = samples_x.permute(0, 2, 1, 3, 4).contiguous()                     # :493
        samples = torch.clamp((samples_x + 1.0) / 2.0, min=0.0, max=1.0)              # :494
    assert samples.shape == (1, FRAMES, 3, target_H, target_W), samples.shape
    np.savez_compressed(
        os.path.join(OUT, "e2e_tiny.npz"), dit_seed=DIT_SEED, vae_seed=VAE_SEED, noise_seed=NOISE_SEED, scale_factor=SCALE_FACTOR,
        steps=3, size=np.array(SIZE), ref_u8=ref_u8.numpy(), pose_u8=pose_u8.numpy(), ctx=ctx.numpy(), uc_ctx=uc_ctx.numpy(),
        clip=clip.numpy(), ref_concat=ref_concat.numpy(), smpl_render_latent=smpl_render_latent.numpy(),
        samples_z=samples_z.numpy(),
        samples=samples.numpy().astype(np.float16))            # [0, 1] video, stored to 2^-11 (the bound of the test is 3e-2)
    print("e2e_tiny: ref_concat", tuple(ref_concat.shape), "pose latent", tuple(pose_latent.shape), "z", tuple(samples_z.shape),
          float(samples_z.abs().mean()), "video", tuple(samples.shape), float(samples.mean()),
          "clamped fraction", float(((samples == 0) | (samples == 1)).float().mean()))


if __name__ == "__main__":
    main()

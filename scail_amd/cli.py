"""Driver mirroring the hot-path part of the reference CLI (sample_video.py:219-507):

    python -m scail_amd.cli --base <model.yaml> [<sampling.yaml>] [--load DIR] [--steps N] [--out out.pt]
    python -m scail_amd.cli --tiny          # BASELINE.json configs[0] shape: 2-layer / 128-dim DiT, 4x8x8 latent, 2 steps

Order of work per request (same as the reference): VAE-encode the reference frame and the half-resolution
pose video (:355-391), build c / uc (:433-470), ``engine.sample`` (:476-483), VAE-decode (:491-494).
The reference's first VAE encode of [ref + zeros] (:362-365) is skipped: ``concat_images`` only gates a
branch and is never read by the network -- a zero-size placeholder is passed.

Offline limits of this image: the UMT5 / CLIP encoders and mp4 decoding (decord, imageio, cv2) are not
available, so requests are tensors: ``--inputs file.pt`` with keys ref (3,1,H,W) in [-1,1], pose (3,T,H,W),
context (1,Lt,4096), uncond_context (1,Lt,4096), clip (1,257,1280); without it synthetic inputs are drawn."""
from __future__ import annotations

import argparse
import os
import time

# the host driver supports only dmabuf IPC: RCCL between the ranks of one node fails without this (set before HIP starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

from . import lib
from .config import load_yaml_configs
from .engine import SATVideoDiffusionEngine

TINY = {
    "model": {
        "use_i2v_clip": True, "scale_factor": 1.0, "build_first_stage": True,
        "network_config": {"target": "dit_video_crossattn_sc_xc.DiffusionTransformer", "params": dict(
            time_freq_dim=256, time_embed_dim=128, share_adaln=True, elementwise_affine=False, num_frames=13,
            time_compressed_rate=4, latent_width=32, latent_height=32, num_layers=2, patch_size=[1, 2, 2], in_channels=20,
            out_channels=16, text_dim=64, hidden_size=128, inner_hidden_size=256, num_attention_heads=1,
            transformer_args=dict(model_parallel_size=1, is_decoder=True),
            modules={"pos_embed_config": {"params": {"hidden_size_head": 128, "interleaved_rope": True}},
                     "adaln_layer_config": {"params": {"qk_ln": True, "hidden_size_head": 128}}})},
        "first_stage_config": {"target": "sgm.models.wan_vae.WanVAE", "params": {"vae_pth": None, "dtype": "torch.bfloat16", "dim": 32}},
        "sampler_config": {"target": "sgm.modules.diffusionmodules.sampling.RFSampler", "params": dict(
            hunyuan_schedule=True, shift_scale=5, num_steps=2,
            guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 4}})},
    },
    "args": {"sampling_image_size": [64, 64], "sampling_fps": 16},
}


def synthetic_request(H, W, frames, text_dim, Lt, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    ctx = r(1, Lt, text_dim)
    ctx[:, Lt // 8:] = 0
    uc = torch.zeros(1, Lt, text_dim)
    uc[:, :1] = r(1, 1, text_dim)
    return dict(ref=(torch.rand(3, 1, H, W, generator=g) * 2 - 1).to(device), pose=(torch.rand(3, frames, H // 2, W // 2, generator=g) * 2 - 1).to(device),
                context=ctx.to(device), uncond_context=uc.to(device), clip=r(1, 257, 1280).to(device))


def run(cfg, inputs=None, steps=None, load=None, seed=1234, device="cuda", frames=None):
    lib.load()
    mc = dict(cfg["model"])
    mc["build_first_stage"] = True
    engine = SATVideoDiffusionEngine(mc, device=device)
    if load:
        from .checkpoint import load_checkpoint
        load_checkpoint(engine, load, force_inference=cfg.get("args", {}).get("force_inference", True))
    H, W = cfg.get("args", {}).get("sampling_image_size", [512, 896])
    net = engine.network
    frames = frames or min(81, net.num_frames)
    req = inputs or synthetic_request(H, W, frames, net.text_dim, 512 if net.text_dim == 4096 else 12, device, seed)
    vae = engine.first_stage_model
    t0 = time.perf_counter()
    ref_lat = vae.encode([req["ref"]])
    pose_lat = vae.encode([req["pose"]])                                    # already half resolution (sample_video.py:350-351)
    ref_concat = ref_lat.permute(0, 2, 1, 3, 4).contiguous().to(torch.bfloat16)      # B C T H W -> B T C H W
    pose_latent = pose_lat.permute(0, 2, 1, 3, 4).contiguous().to(torch.bfloat16)
    T, C, h, w = pose_latent.shape[1], ref_concat.shape[2], ref_concat.shape[3], ref_concat.shape[4]
    shared = dict(concat_images=torch.zeros(1, device=device), ref_concat=ref_concat, concat_pose=pose_latent,
                  concat_smpl_render=pose_latent, image_clip_features=req["clip"].to(torch.bfloat16))
    c = dict(crossattn=req["context"], **shared)
    uc = dict(crossattn=req["uncond_context"], **shared)
    torch.manual_seed(seed)
    z = engine.sample(c, uc=uc, batch_size=1, shape=(T, C, h, w), num_steps=steps)
    z = z.permute(0, 2, 1, 3, 4).contiguous()                               # B T C H W -> B C T H W (:484-485)
    x = engine.decode_first_stage(z.float())
    video = torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)                          # (:494)
    torch.cuda.synchronize()
    return video, z, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", nargs="*", default=[])
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--inputs", default=None)
    ap.add_argument("--load", default=None)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cfg = TINY if a.tiny or not a.base else load_yaml_configs(*a.base)
    inputs = torch.load(a.inputs) if a.inputs else None
    video, z, dt = run(cfg, inputs, a.steps, a.load, a.seed)
    print(f"sampled latent {tuple(z.shape)} -> video {tuple(video.shape)} in {dt:.2f} s")
    if a.out:
        torch.save({"video": video.cpu(), "latent": z.cpu()}, a.out)


if __name__ == "__main__":
    main()

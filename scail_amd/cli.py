"""Driver mirroring the hot-path part of the reference CLI (sample_video.py:219-507):

    python -m scail_amd.cli --base <model.yaml> [<sampling.yaml>] [--load DIR] [--steps N] [--out out.pt]
    python -m scail_amd.cli --tiny          # BASELINE.json configs[0] shape: 2-layer / 128-dim DiT, 4x8x8 latent, 2 steps
    python -m scail_amd.cli --base ... --request "the girl is dancing@@examples/001" [--input-file requests.txt]

Order of work per request (same as the reference): VAE-encode the reference frame and the half-resolution
pose video (:355-391), build c / uc (:433-470), ``engine.sample`` (:476-483), VAE-decode (:491-494).
The reference's first VAE encode of [ref + zeros] (:362-365) is skipped: ``concat_images`` only gates a
branch and is never read by the network -- a zero-size placeholder is passed.

Requests: ``--ref-image ref.jpg --pose-video <frames dir | .npy | animated .webp/.png/.gif | Motion-JPEG .mp4> [--conditioning c.pt]`` runs
the reference's preprocessing (centre crop, [-1, 1], half-resolution pose; scail_amd/preprocess.py) on files, or
``--inputs file.pt`` passes tensors directly: ref (3,1,H,W) in [-1,1], pose (3,T,H,W), context (1,Lt,4096),
uncond_context (1,Lt,4096), clip (1,257,1280); without either synthetic inputs are drawn.  ``--save-dir`` writes
``0_output_000000.webp`` (lossless animated WebP; ``--format`` for APNG / GIF / .npy / frames, or ``.mp4`` = the reference's
file name with Motion-JPEG samples, container written by scail_amd/video_io.py) where the reference writes H.264 mp4.  ``--prompt TEXT --tokenizer <HF dir | spiece.model> [--t5-ckpt ..] [--clip-ckpt ..]`` runs the UMT5 and CLIP encoders
(scail_amd/umt5.py, clip.py) on the prompt and the reference image.  Offline limits of this image: no H.264 / HEVC codec (decord,
imageio, ffmpeg, cv2 are absent: an .mp4 with such a track is rejected by the name of its codec), no tokenizer files and no checkpoints -- hence the container formats above, the tokenizer as a path
argument, and random-init weights unless checkpoints are given."""
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


REF_IMAGE_PATTERNS = ["ref.jpg", "ref.png", "ref_image.jpg", "ref_image.png"]                # sample_video.py:289
# the reference looks for rendered_aligned.mp4 / rendered.mp4 (:296); the containers this image can decode come first
POSE_PATTERNS = [stem + ext for stem in ("rendered_aligned", "rendered") for ext in ("", ".webp", ".png", ".gif", ".npy", ".pt", ".mp4")]      # (.mp4 last: the reference's own examples are H.264, which nothing here decodes)


def find_file_with_patterns(directory: str, patterns):
    """sample_video.py:64-70."""
    import os
    for pat in patterns:
        p = os.path.join(directory, pat)
        if os.path.exists(p):
            return p
    return None


def parse_request(line: str):
    """One request line of the reference CLI, ``<prompt>@@<example_dir>`` (sample_video.py:76, :284-300): the directory holds
    the reference image and the rendered pose video.  -> (prompt, example_dir, image_path, pose_path)."""
    parts = line.strip().split("@@")
    if len(parts) != 2:
        raise ValueError(f"expected '<prompt>@@<example_dir>', got {line!r}")
    text, input_dir = parts[0], parts[1]
    if text == "None":
        text = ""
    image_path = find_file_with_patterns(input_dir, REF_IMAGE_PATTERNS)
    if image_path is None:
        raise FileNotFoundError(f"Reference image not found in {input_dir}. Tried: {REF_IMAGE_PATTERNS}")
    pose_path = find_file_with_patterns(input_dir, POSE_PATTERNS)
    if pose_path is None:
        raise FileNotFoundError(f"Pose video not found in {input_dir}. Tried: {POSE_PATTERNS}")
    return text, input_dir, image_path, pose_path


def read_from_file(path: str, rank: int = 0, world_size: int = 1):
    """sample_video.py:82-91: request lines of a text file, dealt round-robin to the data-parallel ranks."""
    with open(path) as f:
        for cnt, line in enumerate(f):
            if cnt % world_size == rank and line.strip():
                yield line.strip(), cnt


def encode_conditioning(prompt: str, negative_prompt: str, ref: torch.Tensor, text_dim: int, tokenizer_path: str,
                        t5_ckpt: str = None, clip_ckpt: str = None, device="cuda", max_length: int = 512):
    """Prompt + reference image -> the conditioning tensors of the request (sample_video.py:397-400, :416-438): UMT5
    states of the prompt and of the negative prompt (padded rows zeroed), CLIP ViT-H penultimate features of the reference
    frame.  Both encoders run once and are released (the reference moves them back to the CPU).  Checkpoints are optional
    (random init without them -- there are none offline); a text width other than 4096 builds a 2-layer encoder of that
    width, for plumbing tests against small networks."""
    from .clip import CLIPModel
    from .tokenizer import HuggingfaceTokenizer
    from .umt5 import T5EncoderModel
    kw = {}
    if text_dim != 4096:
        if text_dim % 128:
            raise ValueError("the plumbing-size text encoder needs a text width that is a multiple of 128")
        vocab = HuggingfaceTokenizer(tokenizer_path).vocab_size
        kw = dict(vocab=(vocab + 63) // 64 * 64, dim=text_dim, dim_attn=text_dim, dim_ffn=2 * text_dim,
                  num_heads=max(1, text_dim // 64), num_layers=2)
    t5 = T5EncoderModel(max_length=max_length, checkpoint_path=t5_ckpt, device=device, tokenizer_path=tokenizer_path, **kw)
    ctx = t5.encode_text([prompt, negative_prompt])
    del t5
    clip = CLIPModel(device=device, checkpoint_path=clip_ckpt)
    feats = clip.visual([ref.to(device)])                                               # (1, 257, 1280)
    del clip
    torch.cuda.empty_cache()
    return dict(context=ctx[0:1].contiguous(), uncond_context=ctx[1:2].contiguous(), clip=feats)


def request_from_files(ref_image: str, pose_video: str, cfg, conditioning: str = None, device="cuda", seed=0, text_dim=4096):
    """The reference's request assembly from files (sample_video.py:300-351): reference image + driving (pose) video ->
    centre-cropped, [-1, 1], pose at half resolution (``smpl_downsample``).  Text / CLIP conditioning comes from
    ``conditioning`` (a .pt with context, uncond_context, clip) -- the T5 tokenizer files are not available offline --
    or is drawn synthetically."""
    from . import preprocess, video_io
    img = video_io.load_image_to_tensor_chw_normalized(ref_image)                       # (1, 3, H, W) in [-1, 1]
    H, W = preprocess.target_size((img.shape[2], img.shape[3]), cfg.get("args", {}).get("sampling_image_size", [512, 896]))
    img = preprocess.prepare_reference_image(img, (H, W))
    pose = video_io.load_video_for_pose_sample(pose_video).permute(0, 3, 1, 2)         # T H W C -> T C H W (:339)
    _, smpl = preprocess.prepare_pose_video(pose, (H, W), downsample=True)
    req = synthetic_request(H, W, smpl.shape[0], text_dim, 512 if text_dim == 4096 else 12, device, seed)
    if conditioning:
        req.update({k: v.to(device) for k, v in torch.load(conditioning, map_location="cpu").items()
                    if k in ("context", "uncond_context", "clip")})
    req["ref"] = img.permute(1, 0, 2, 3).contiguous().to(device)                        # (3, 1, H, W)
    req["pose"] = smpl.permute(1, 0, 2, 3).contiguous().to(device)                      # (3, T, H/2, W/2)
    return req, (H, W)


def build_engine(cfg, load=None, device="cuda"):
    """Under ``python -m torch.distributed.run`` the whole world is one sequence-parallel group (the reference's CLI: dp = 1,
    sample_video.py:229): every rank builds the engine, encodes the request and joins ``engine.sample``; SP rank 0 decodes
    and saves (:484-507)."""
    lib.load()
    mc = dict(cfg["model"])
    mc["build_first_stage"] = True
    sp = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from . import parallel
        backend = os.environ.get("SCAIL_DIST_BACKEND", "nccl")          # "gloo": ranks sharing one GPU (tests only)
        if backend == "nccl":
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
        sp = parallel.init_from_env(backend)
    engine = SATVideoDiffusionEngine(mc, device=device, sp=sp)
    if load:
        from .checkpoint import load_checkpoint
        load_checkpoint(engine, load, force_inference=cfg.get("args", {}).get("force_inference", True))
    return engine


def run(cfg, inputs=None, steps=None, load=None, seed=1234, device="cuda", frames=None, engine=None):
    engine = engine or build_engine(cfg, load, device)
    H, W = cfg.get("args", {}).get("sampling_image_size", [512, 896])
    net = engine.network
    frames = frames or min(81, net.num_frames)
    if callable(inputs):                                      # file-based request: needs the network's text width
        inputs = inputs(net.text_dim)
    req = inputs or synthetic_request(H, W, frames, net.text_dim, 512 if net.text_dim == 4096 else 12, device, seed)
    t0 = time.perf_counter()
    # model.encode_first_stage(..., force_encode=True): VAE mean x scale_factor (sample_video.py:366, :381; diffusion_video.py:311-331)
    ref_lat = engine.encode_first_stage(req["ref"].unsqueeze(0), None, force_encode=True)
    pose_lat = engine.encode_first_stage(req["pose"].unsqueeze(0), None, force_encode=True)   # already half resolution (:350-351)
    ref_concat = ref_lat.permute(0, 2, 1, 3, 4).contiguous().to(torch.bfloat16)      # B C T H W -> B T C H W
    pose_latent = pose_lat.permute(0, 2, 1, 3, 4).contiguous().to(torch.bfloat16)
    T, C, h, w = pose_latent.shape[1], ref_concat.shape[2], ref_concat.shape[3], ref_concat.shape[4]
    shared = dict(concat_images=torch.zeros(1, device=device), ref_concat=ref_concat, concat_pose=pose_latent,
                  concat_smpl_render=pose_latent, image_clip_features=req["clip"].to(torch.bfloat16))
    c = dict(crossattn=req["context"], **shared)
    uc = dict(crossattn=req["uncond_context"], **shared)
    torch.manual_seed(seed)
    z = engine.sample(c, uc=uc, batch_size=1, shape=(T, C, h, w), num_steps=steps)
    if engine.sp is not None and engine.sp.size > 1 and engine.sp.rank != 0:
        torch.cuda.synchronize()
        return None, None, time.perf_counter() - t0                         # only SP rank 0 holds the gathered latent (:484)
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
    ap.add_argument("--ref-image", default=None, help="reference image file (any Pillow format)")
    ap.add_argument("--pose-video", default=None, help="driving video: directory of frames, .npy/.pt (T,H,W,3) or animated WebP/PNG/GIF")
    ap.add_argument("--conditioning", default=None, help=".pt with context / uncond_context / clip tensors")
    ap.add_argument("--prompt", default=None, help="text prompt (needs --tokenizer; UMT5 / CLIP run on the GPU, random-init without --t5-ckpt / --clip-ckpt)")
    ap.add_argument("--negative-prompt", default="")
    ap.add_argument("--tokenizer", default=None, help="Hugging Face tokenizer directory or SentencePiece .model file of umt5-xxl")
    ap.add_argument("--t5-ckpt", default=None)
    ap.add_argument("--clip-ckpt", default=None)
    ap.add_argument("--save-dir", default=None, help="write <key>_000000.<ext> like the reference's save_multi_video_grid_and_mp4")
    ap.add_argument("--format", default=".webp", help=".webp (lossless) | .png (APNG) | .gif | .npy | .mp4 (Motion JPEG) | '' (directory of PNG frames)")
    ap.add_argument("--request", action="append", default=[], help="'<prompt>@@<example_dir>' (the reference's cli input line); repeatable")
    ap.add_argument("--input-file", default=None, help="text file of request lines (the reference's --input-type txt)")
    ap.add_argument("--output-dir", default="outputs", help="results of --request / --input-file go to <output-dir>/<example>/")
    a = ap.parse_args()
    cfg = TINY if a.tiny or not a.base else load_yaml_configs(*a.base)
    lines = [(r, i) for i, r in enumerate(a.request)] + (list(read_from_file(a.input_file)) if a.input_file else [])
    if lines:
        import os
        from . import video_io
        engine = build_engine(cfg, a.load)
        td = engine.network.text_dim
        for line, cnt in lines:
            text, input_dir, image_path, pose_path = parse_request(line)
            print(cnt, ": ", text)
            req = request_from_files(image_path, pose_path, cfg, a.conditioning, seed=a.seed, text_dim=td)[0]
            if a.tokenizer:
                req.update(encode_conditioning(text, a.negative_prompt, req["ref"], td, a.tokenizer, a.t5_ckpt, a.clip_ckpt,
                                               max_length=512 if td == 4096 else 16))
            video, z, dt = run(cfg, req, a.steps, seed=a.seed, engine=engine)
            if video is None:
                continue
            save_dir = os.path.join(a.output_dir, os.path.basename(os.path.normpath(input_dir)))
            os.makedirs(save_dir, exist_ok=True)
            with open(os.path.join(save_dir, "text.txt"), "w") as f:                   # sample_video.py:413-414
                f.write(text)
            samples = video.permute(0, 2, 1, 3, 4).contiguous().cpu()
            paths = video_io.save_multi_video_grid([samples], save_dir, fps=cfg.get("args", {}).get("sampling_fps", 16),
                                                   key=f"{os.path.basename(os.path.normpath(input_dir))}_output", ext=a.format)
            print(f"  latent {tuple(z.shape)} -> video {tuple(video.shape)} in {dt:.2f} s; wrote {', '.join(paths)}")
        return
    inputs = torch.load(a.inputs) if a.inputs else None
    if a.ref_image or a.pose_video:
        if not (a.ref_image and a.pose_video):
            ap.error("--ref-image and --pose-video go together")
        def inputs(text_dim):
            req = request_from_files(a.ref_image, a.pose_video, cfg, a.conditioning, seed=a.seed, text_dim=text_dim)[0]
            if a.prompt is not None:
                if not a.tokenizer:
                    ap.error("--prompt needs --tokenizer (tokenizer files are not bundled)")
                req.update(encode_conditioning(a.prompt, a.negative_prompt, req["ref"], text_dim, a.tokenizer, a.t5_ckpt, a.clip_ckpt,
                                               max_length=512 if text_dim == 4096 else 16))
            return req
    video, z, dt = run(cfg, inputs, a.steps, a.load, a.seed)
    if video is None:
        return
    print(f"sampled latent {tuple(z.shape)} -> video {tuple(video.shape)} in {dt:.2f} s")
    if a.out:
        torch.save({"video": video.cpu(), "latent": z.cpu()}, a.out)
    if a.save_dir:
        from . import video_io
        samples = video.permute(0, 2, 1, 3, 4).contiguous().cpu()                      # B C T H W -> B T C H W (:493)
        paths = video_io.save_multi_video_grid([samples], a.save_dir, fps=cfg.get("args", {}).get("sampling_fps", 16),
                                               key="0_output", ext=a.format)
        print("wrote", ", ".join(paths))


if __name__ == "__main__":
    main()

"""File IO either side of the sampling path (reference sample_video.py:181-217 writers, :300-351 readers).

The reference decodes the driving video with decord and writes mp4 through imageio/ffmpeg; neither exists in this
image (no codecs offline), so the same roles are filled with what Pillow can do losslessly or near-losslessly:

    read   reference image      any Pillow format                        -> load_image_to_tensor_chw_normalized
           driving / GT video   directory of frames, .npy / .pt arrays,
                                animated WebP / PNG (APNG) / GIF         -> load_video_for_pose_sample
    write  result video         animated WebP (default; lossless), APNG,
                                GIF, .npy, or a directory of PNG frames  -> save_video_as_grid / save_multi_video_grid

Tensor conventions are the reference's: readers return uint8 (T, H, W, C) like ``load_video_for_pose_sample`` /
a [-1, 1] (1, C, H, W) image; writers take (B, T, C, H, W) in [0, 1] and quantise with ``(255 * x).astype(uint8)``
(truncation, :188 / :212).  A request for .mp4 raises with the reason instead of silently writing something else.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np
import torch

_ANIMATED = (".webp", ".png", ".apng", ".gif")
_NO_CODEC = (".mp4", ".mov", ".mkv", ".avi", ".webm")


def _pil():
    from PIL import Image, ImageSequence
    return Image, ImageSequence


def load_image_to_tensor_chw_normalized(path: str) -> torch.Tensor:
    """(1, 3, H, W) float32 in [-1, 1] (reference helper of the same name, sample_video.py:343 context)."""
    Image, _ = _pil()
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.float32)
    t = torch.from_numpy(arr).permute(2, 0, 1).unsqueeze(0)
    return (t - 127.5) / 127.5


def load_video_for_pose_sample(path: str, max_frames: int = None) -> torch.Tensor:
    """(T, H, W, 3) uint8 (what the decord-based reference helper of the same name returns, sample_video.py:48-54)."""
    ext = os.path.splitext(path)[1].lower()
    if os.path.isdir(path):
        Image, _ = _pil()
        names = sorted(n for n in os.listdir(path) if os.path.splitext(n)[1].lower() in (".png", ".jpg", ".jpeg", ".bmp", ".webp"))
        if not names:
            raise FileNotFoundError(f"no image frames in {path}")
        frames = []
        for n in names[:max_frames]:
            with Image.open(os.path.join(path, n)) as im:
                frames.append(np.asarray(im.convert("RGB")))
        arr = np.stack(frames)
    elif ext == ".npy":
        arr = np.load(path)
    elif ext in (".pt", ".pth"):
        arr = torch.load(path, map_location="cpu").numpy()
    elif ext in _ANIMATED:
        Image, ImageSequence = _pil()
        with Image.open(path) as im:
            arr = np.stack([np.asarray(f.convert("RGB")) for f in ImageSequence.Iterator(im)])
    elif ext in _NO_CODEC:
        raise RuntimeError(f"{path}: no video decoder in this environment (decord / ffmpeg are absent offline); "
                           "pass a directory of frames, an .npy / .pt array (T,H,W,3) or an animated WebP / PNG / GIF")
    else:
        raise ValueError(f"unsupported driving-video input {path}")
    if arr.ndim != 4 or arr.shape[-1] != 3:
        raise ValueError(f"{path}: expected (T, H, W, 3), got {arr.shape}")
    if max_frames is not None:
        arr = arr[:max_frames]
    if arr.dtype != np.uint8:
        arr = np.clip(np.rint(arr), 0, 255).astype(np.uint8)
    return torch.from_numpy(np.ascontiguousarray(arr))


def _to_u8_frames(vid: torch.Tensor) -> List[np.ndarray]:
    """vid (T, C, H, W) in [0, 1] -> list of (H, W, C) uint8, quantised like the reference (:211-212)."""
    return [(255.0 * f.permute(1, 2, 0)).cpu().numpy().astype(np.uint8) for f in vid]


def _write_frames(frames: Sequence[np.ndarray], path: str, fps: float):
    ext = os.path.splitext(path)[1].lower()
    if ext in _NO_CODEC:
        raise RuntimeError(f"{path}: no video encoder in this environment (imageio / ffmpeg are absent offline); "
                           "use .webp (lossless, default), .png (APNG), .gif, .npy or a directory")
    if ext == ".npy":
        np.save(path, np.stack(frames))
        return
    Image, _ = _pil()
    ims = [Image.fromarray(f) for f in frames]
    dur = max(1, int(round(1000.0 / float(fps))))
    if ext == "":
        os.makedirs(path, exist_ok=True)
        for i, im in enumerate(ims):
            im.save(os.path.join(path, f"{i:06d}.png"))
    elif ext == ".webp":
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=dur, loop=0, lossless=True, method=0)
    elif ext in (".png", ".apng"):
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=dur, loop=0, format="PNG")
    elif ext == ".gif":
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=dur, loop=0)
    else:
        raise ValueError(f"unsupported output format {ext}")


def save_video_as_grid(video_batch: torch.Tensor, save_path: str, fps: float = 5, ext: str = ".webp") -> List[str]:
    """video_batch (B, T, C, H, W) in [0, 1] -> save_path/000000<ext>, ... (reference save_video_as_grid_and_mp4, :201-217)."""
    os.makedirs(save_path, exist_ok=True)
    out = []
    for i, vid in enumerate(video_batch):
        p = os.path.join(save_path, f"{i:06d}{ext}")
        _write_frames(_to_u8_frames(vid), p, fps)
        out.append(p)
    return out


def save_multi_video_grid(video_batches: Sequence[torch.Tensor], save_dir: str, fps: float = 5, key: str = "0_output",
                          ext: str = ".webp") -> List[str]:
    """Several (B, T, C, H, W) clips side by side, frame layout "n c h w -> h (n w) c"
    (reference save_multi_video_grid_and_mp4, :181-198): save_dir/<key>_000000<ext>."""
    os.makedirs(save_dir, exist_ok=True)
    multi = torch.stack([v.float().cpu() for v in video_batches], dim=2)           # B T N C H W
    out = []
    for i, mv in enumerate(multi):
        frames = []
        for fr in mv:                                                              # N C H W
            n, c, h, w = fr.shape
            grid = fr.permute(2, 0, 3, 1).reshape(h, n * w, c)
            frames.append((255.0 * grid).numpy().astype(np.uint8))
        p = os.path.join(save_dir, f"{key}_{i:06d}{ext}")
        _write_frames(frames, p, fps)
        out.append(p)
    return out

"""Request preprocessing of the reference CLI (sample_video.py:325-351, data_video.py:141-170), as tensor ops.

    resize_for_rectangle_crop   data_video.py:141-170: scale so the frame covers the target, centre crop
    prepare_pose_video          sample_video.py:340-351: crop, (x - 127.5) / 127.5, optional 0.5x bilinear
    prepare_reference_image     sample_video.py:343 (+ the [-1, 1] normalisation of the image loader)
    target_size                 sample_video.py:325-328: sampling_image_size is (H, W) for landscape, swapped for portrait

The reference resizes with torchvision ``resize(..., BICUBIC)`` on uint8 tensors; torchvision is not available
offline, so the resize is restated with ``torch.nn.functional.interpolate(mode="bicubic", antialias=True)`` followed
by the uint8 round + clamp torchvision applies (its tensor path calls exactly that op) -- parity for this module is
UNPINNED (no reference output could be generated here); the geometry (sizes, crop offsets, value ranges) is tested.
Video decoding (decord) and mp4 writing (imageio) need packages that are absent offline; ``scail_amd/video_io.py`` fills
those roles with Pillow containers (frame directories, arrays, animated WebP / PNG / GIF)."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch
import torch.nn.functional as F


def target_size(image_hw: Tuple[int, int], sampling_image_size: Sequence[int]) -> Tuple[int, int]:
    """(target_H, target_W): sample_video.py:325-328."""
    h, w = image_hw
    a, b = sampling_image_size
    return (a, b) if h < w else (b, a)


def _resize_bicubic_u8(arr: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    dt = arr.dtype
    out = F.interpolate(arr.float(), size=size, mode="bicubic", align_corners=False, antialias=True)
    if dt == torch.uint8:
        out = out.round().clamp_(0, 255).to(torch.uint8)
    return out


def resize_for_rectangle_crop(arr: torch.Tensor, image_size: Sequence[int], reshape_mode: str = "center") -> torch.Tensor:
    """arr (T, C, H, W).  data_video.py:141-170; only the deterministic 'center' mode the CLI uses."""
    H, W = arr.shape[2], arr.shape[3]
    th, tw = int(image_size[0]), int(image_size[1])
    if W / H > tw / th:
        arr = _resize_bicubic_u8(arr, (th, int(W * th / H)))
    else:
        arr = _resize_bicubic_u8(arr, (int(H * tw / W), tw))
    h, w = arr.shape[2], arr.shape[3]
    if reshape_mode != "center":
        raise NotImplementedError("only reshape_mode='center' (the sampling CLI); 'random' is a training augmentation")
    top, left = (h - th) // 2, (w - tw) // 2
    return arr[:, :, top:top + th, left:left + tw]


def prepare_pose_video(pose_u8: torch.Tensor, size_hw: Sequence[int], downsample: bool = True):
    """pose_u8 (T, C, H, W) 0..255 -> (pose [-1,1] at full size, smpl render at half size if ``downsample``),
    sample_video.py:340-351."""
    pose = resize_for_rectangle_crop(pose_u8, size_hw, "center").float()
    pose = (pose - 127.5) / 127.5
    smpl = F.interpolate(pose, scale_factor=0.5, mode="bilinear", align_corners=False) if downsample else pose
    return pose, smpl


def prepare_reference_image(img: torch.Tensor, size_hw: Sequence[int]) -> torch.Tensor:
    """img (1, C, H, W) already in [-1, 1] (the reference's loader normalises) -> centre-cropped to size."""
    return resize_for_rectangle_crop(img, size_hw, "center")

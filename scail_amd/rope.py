"""Host-side construction of the 3-segment 3D RoPE tables consumed by ``scail_rmsnorm_rope``.

Mirrors Rotary3DPositionEmbeddingMixin (reference dit_video_crossattn_sc_xc.py:404-513 for the
frequency grids, :525-645 for the three per-segment slicing rules).  Built once per (latent shape,
sequence-parallel shift) on the CPU in fp32 exactly like the reference builds its buffers, then
uploaded; only the (L, head_dim/2) pair tables are kept because the interleaved layout repeats
every angle twice (``repeat(..., '... n -> ... (n r)', r=2)``, :449-451) -- also after the 2x2
average pooling of the pose segment, which acts per channel.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F


def rope_dims(head_dim: int) -> Tuple[int, int, int]:
    """(dim_t, dim_h, dim_w) -- reference :404-406 (128 -> 44/42/42)."""
    dim_t = head_dim - 4 * (head_dim // 6)
    dim_h = (head_dim // 6) * 2
    return dim_t, dim_h, dim_h


def _freqs(dim: int, theta: float) -> torch.Tensor:
    return 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))


def _pair_angles(head_dim, theta, t_pos, h_pos, w_pos):
    """(T,H,W,head_dim/2) angles, one per interleaved pair."""
    dt, dh, dw = rope_dims(head_dim)
    ft = torch.einsum("p,f->pf", t_pos.float(), _freqs(dt, theta))
    fh = torch.einsum("p,f->pf", h_pos.float(), _freqs(dh, theta))
    fw = torch.einsum("p,f->pf", w_pos.float(), _freqs(dw, theta))
    T, H, W = ft.shape[0], fh.shape[0], fw.shape[0]
    return torch.cat([ft[:, None, None, :].expand(T, H, W, -1), fh[None, :, None, :].expand(T, H, W, -1),
                      fw[None, None, :, :].expand(T, H, W, -1)], dim=-1)


def build_tables(head_dim: int, rope_T: int, rope_H: int, rope_W: int, H_shift: int = 0, W_shift: int = 0,
                 global_rope_H: int = 0, global_rope_W: int = 120, theta: float = 10000.0,
                 max_T: int = None, max_H: int = None, max_W: int = None, n_char: int = 1):
    """cos, sin: (L, head_dim/2) fp32 CPU tensors for the token order [ref | noise | pose].

    noise: t = 1..T (grid_t :424), (h, w) = shift + index            (rotary      :543-551)
    ref:   t = 0     (grid_extended_t :428), same (h, w) window       (rotary_ref  :579-588)
    pose:  window offset by (global_rope_H, global_rope_W)=(0,120) in the noise tables, then
           avg_pool2d(2) of cos and sin separately                    (rotary_pose :616-637)
    ``max_*`` (table extents of the reference: T=(num_frames-1)//4+1, H=latent_height//2,
    W=latent_width//2 + 120) are only used to reject windows the reference could not index.

    n_char > 1 is an EXTENSION (BASELINE config 5; the reference has exactly one reference frame and one pose stream,
    :1559): token order [ref_0..ref_{C-1} | noise | pose_0..pose_{C-1}]; character k > 0 takes windows the reference
    leaves unused -- ref_k at t = 0 with the w window shifted by k * global_rope_W, pose_k at t = 1..T with the w window
    shifted by global_rope_W + k * rope_W.  n_char == 1 is exactly the reference.
    """
    if max_T is not None and rope_T > max_T:
        raise ValueError(f"rope_T {rope_T} exceeds the table extent {max_T}")
    if max_H is not None and max(H_shift, global_rope_H + H_shift) + rope_H > max_H:
        raise ValueError("RoPE H window exceeds the table extent")
    if n_char < 1:
        raise ValueError("n_char must be >= 1")
    if max_W is not None and max(global_rope_W + W_shift + n_char * rope_W, (n_char - 1) * global_rope_W + W_shift + rope_W) > max_W:
        raise ValueError("RoPE W window exceeds the table extent")
    if rope_H % 2 or rope_W % 2:
        raise ValueError("pose tokens need even patch-grid extents (2x2 pooling)")
    hp = torch.arange(H_shift, H_shift + rope_H)
    wp = torch.arange(W_shift, W_shift + rope_W)
    a_noise = _pair_angles(head_dim, theta, torch.arange(1, rope_T + 1), hp, wp)
    hp2 = torch.arange(global_rope_H + H_shift, global_rope_H + H_shift + rope_H)
    wp2 = torch.arange(global_rope_W + W_shift, global_rope_W + W_shift + rope_W)

    def pool(x):
        return F.avg_pool2d(x.permute(0, 3, 1, 2), kernel_size=2, stride=2).permute(0, 2, 3, 1)

    half = head_dim // 2
    segs = [_pair_angles(head_dim, theta, torch.tensor([0]), hp, wp + k * global_rope_W) for k in range(n_char)]
    segs.append(a_noise)
    cos = [a.cos().reshape(-1, half) for a in segs]
    sin = [a.sin().reshape(-1, half) for a in segs]
    for k in range(n_char):
        a_pose = _pair_angles(head_dim, theta, torch.arange(1, rope_T + 1), hp2, wp2 + k * rope_W)
        cos.append(pool(a_pose.cos()).reshape(-1, half))
        sin.append(pool(a_pose.sin()).reshape(-1, half))
    return torch.cat(cos, dim=0).contiguous(), torch.cat(sin, dim=0).contiguous()

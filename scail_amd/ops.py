"""Tensor-level wrappers over the C ABI (include/scail_hip.h).

PyTorch is used for device memory and the current HIP stream only; every function enqueues one
HIP kernel of libscail_hip.so on ``torch.cuda.current_stream()``.  All tensors must live on the
GPU -- there is no CPU path.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import lib as L

bf16 = torch.bfloat16
f32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise L.ScailHipError(f"{name}: tensor must be on the GPU (scail_amd has no CPU path)")
    if t.dtype != dtype:
        raise L.ScailHipError(f"{name}: expected {dtype}, got {t.dtype}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rowmajor2d(t: torch.Tensor, name: str):
    """(rows, cols, row_stride) of a tensor whose last dim is contiguous and whose leading dims
    collapse to ONE uniformly strided row index (contiguous tensors and column-slice views do)."""
    if t.stride(-1) != 1:
        raise L.ScailHipError(f"{name}: last dim must be contiguous")
    cols = t.shape[-1]
    if t.dim() == 1:
        return 1, cols, cols
    ld = t.stride(-2)
    rows, exp = 1, ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            raise L.ScailHipError(f"{name}: leading dims are not uniformly strided {tuple(t.shape)} {t.stride()}")
        exp *= t.shape[d]
        rows *= t.shape[d]
    return rows, cols, ld


# ------------------------------------------------------------------------------------------------
def gemm(x, w, bias=None, out=None, epilogue=L.EPI_BIAS, resid=None, gate=None, rows_per_batch=0):
    """y = epilogue(x @ w.T + bias).  x (..., K) bf16 (rows may be strided), w (N, K) bf16 contiguous,
    bias fp32 (N).  RESID: y = resid + gate[b] * (.), gate fp32 (n_batch, N) view with row stride."""
    _chk(x, bf16, "gemm.x"); _chk(w, bf16, "gemm.w")
    M, K, lda = _rowmajor2d(x, "gemm.x")
    N = w.shape[0]
    if w.shape[1] != K or not w.is_contiguous():
        raise L.ScailHipError("gemm: w must be contiguous (N, K)")
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=bf16)
    _chk(out, bf16, "gemm.out")
    Mo, No, ldc = _rowmajor2d(out, "gemm.out")
    assert Mo == M and No == N, (Mo, M, No, N)
    ldr = 0
    if resid is not None:
        _chk(resid, bf16, "gemm.resid")
        Mr, Nr, ldr = _rowmajor2d(resid, "gemm.resid")
        assert Mr == M and Nr == N
    gs = 0
    if gate is not None:
        _chk(gate, f32, "gemm.gate")
        assert gate.dim() == 2 and gate.shape[1] == N and gate.stride(1) == 1
        gs = gate.stride(0)
    if bias is not None:
        _chk(bias, f32, "gemm.bias")
    L.call("scail_gemm_bf16", x.data_ptr(), lda, w.data_ptr(), _ptr(bias), out.data_ptr(), ldc, M, N, K,
           epilogue, _ptr(resid), ldr, _ptr(gate), gs, rows_per_batch, _stream())
    return out


def ln_modulate(x, shift, scale, out=None, eps=1e-6, rows_out=None, src_rows_per_batch=None, src_row_offset=0):
    """x (B, Ls, D) bf16; shift/scale fp32 (B, D) views (same row stride).  Output (B, rows_out, D)."""
    _chk(x, bf16, "ln_modulate.x"); _chk(shift, f32, "shift"); _chk(scale, f32, "scale")
    B, Ls, D = x.shape
    assert x.stride(2) == 1 and x.stride(0) == Ls * x.stride(1)
    rows_out = Ls if rows_out is None else rows_out
    src_rows_per_batch = Ls if src_rows_per_batch is None else src_rows_per_batch
    if out is None:
        out = torch.empty(B, rows_out, D, device=x.device, dtype=bf16)
    assert shift.shape == (B, D) and scale.shape == (B, D) and shift.stride(0) == scale.stride(0)
    assert shift.stride(1) == 1 and scale.stride(1) == 1
    L.call("scail_ln_modulate", x.data_ptr(), x.stride(1), out.data_ptr(), out.stride(-2), shift.data_ptr(),
           scale.data_ptr(), shift.stride(0), B, rows_out, src_rows_per_batch, src_row_offset, D, eps, _stream())
    return out


def layernorm_affine(x, w, b, out=None, eps=1e-6):
    _chk(x, bf16, "layernorm_affine.x"); _chk(w, f32, "w"); _chk(b, f32, "b")
    rows, D, ldx = _rowmajor2d(x, "layernorm_affine.x")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=bf16)
    _, _, ldy = _rowmajor2d(out, "layernorm_affine.out")
    L.call("scail_layernorm_affine", x.data_ptr(), ldx, out.data_ptr(), ldy, w.data_ptr(), b.data_ptr(), rows, D,
           eps, _stream())
    return out


ATTN_Q_PRESCALED = -1.0        # include/scail_hip.h SCAIL_ATTN_Q_PRESCALED
ATTN_LOG2_SCALE = 1.4426950408889634 / math.sqrt(128.0)     # q * this = queries in log2 units (flash_attn(..., q_prescaled=True))


def rmsnorm_rope(x, w, cos=None, sin=None, out=None, rows_per_batch=None, head_dim=128, eps=1e-6, out_scale=1.0):
    """RMSNorm over the last dim (+RoPE with (L, head_dim/2) fp32 tables), times ``out_scale`` before the rounding to bf16.
    In place when out is None."""
    _chk(x, bf16, "rmsnorm_rope.x"); _chk(w, f32, "w")
    rows, D, ldx = _rowmajor2d(x, "rmsnorm_rope.x")
    out = x if out is None else out
    _, _, ldy = _rowmajor2d(out, "rmsnorm_rope.out")
    if cos is not None:
        _chk(cos, f32, "cos"); _chk(sin, f32, "sin")
        assert cos.is_contiguous() and sin.is_contiguous() and cos.shape[1] == head_dim // 2
        rows_per_batch = cos.shape[0] if rows_per_batch is None else rows_per_batch
        assert rows_per_batch <= cos.shape[0]
    else:
        rows_per_batch = rows if rows_per_batch is None else rows_per_batch
    L.call("scail_rmsnorm_rope_scaled", x.data_ptr(), ldx, out.data_ptr(), ldy, w.data_ptr(), _ptr(cos), _ptr(sin), rows,
           rows_per_batch, D, head_dim, eps, float(out_scale), _stream())
    return out


def rmsnorm_rope_slabs(x, w, out, cos=None, sin=None, rows_per_batch=None, head_dim=128, eps=1e-6, out_scale=1.0):
    """RMSNorm (+RoPE, x out_scale) of x (rows, D) (row-strided view ok) written as COLUMN SLABS: out (n_slabs, rows, D / n_slabs),
    last dim contiguous, any row / slab stride -- the (destination rank, token, head-group columns) send layout of the Ulysses
    all-to-all; a view ``msg[:, :, j * Dn:(j + 1) * Dn]`` of a (n_slabs, rows, 3 * Dn) message puts q | k | v side by side.  ``w`` None:
    plain copy of x into that layout (include/scail_hip.h scail_rmsnorm_rope_slabs)."""
    _chk(x, bf16, "rmsnorm_rope_slabs.x"); _chk(out, bf16, "rmsnorm_rope_slabs.out")
    rows, D, ldx = _rowmajor2d(x, "rmsnorm_rope_slabs.x")
    if out.dim() != 3 or out.stride(2) != 1 or out.shape[1] != rows or out.shape[0] * out.shape[2] != D:
        raise L.ScailHipError(f"rmsnorm_rope_slabs.out must be a (n_slabs, {rows}, {D} / n_slabs) tensor with a contiguous last dim, got {tuple(out.shape)}")
    if w is not None:
        _chk(w, f32, "w")
    if cos is not None:
        _chk(cos, f32, "cos"); _chk(sin, f32, "sin")
        assert cos.is_contiguous() and sin.is_contiguous() and cos.shape[1] == head_dim // 2
        rows_per_batch = cos.shape[0] if rows_per_batch is None else rows_per_batch
    else:
        rows_per_batch = rows if rows_per_batch is None else rows_per_batch
    L.call("scail_rmsnorm_rope_slabs", x.data_ptr(), ldx, out.data_ptr(), out.shape[2], out.stride(1), out.stride(0), _ptr(w), _ptr(cos),
           _ptr(sin), rows, rows_per_batch, D, head_dim, eps, float(out_scale), _stream())
    return out


def transpose_v(v, heads, head_dim=128, out=None):
    """v (B, Lk, heads*head_dim) bf16 (strided view ok) -> vt (B, heads, head_dim, ceil64(Lk))."""
    _chk(v, bf16, "transpose_v.v")
    B, Lk, D = v.shape
    assert D == heads * head_dim and v.stride(2) == 1
    Lkp = (Lk + 63) // 64 * 64
    if out is None:
        out = torch.empty(B, heads, head_dim, Lkp, device=v.device, dtype=bf16)
    assert out.is_contiguous() and out.shape == (B, heads, head_dim, Lkp)
    L.call("scail_transpose_v", v.data_ptr(), v.stride(1), v.stride(0), out.data_ptr(), B, heads, head_dim, Lk,
           _stream())
    return out


def flash_attn(q, k, vt, out=None, scale=None, accumulate=False, n_seg=1, k_seg_stride=0, vt_seg_stride=0,
               k_broadcast=False, q_prescaled=False):
    """q (B, Lq, H*128) view; k (B|1, Lk, H*128) view [per segment]; vt (B|1, H, 128, Lkp) [per segment].
    Output (B, Lq, H*128) view.  ``k_broadcast``: K/V have batch 1 and are shared by all B queries.
    ``q_prescaled``: q already carries scale * log2(e) (rmsnorm_rope(..., out_scale=ATTN_LOG2_SCALE)); include/scail_hip.h SCAIL_ATTN_Q_PRESCALED."""
    _chk(q, bf16, "flash_attn.q"); _chk(k, bf16, "k"); _chk(vt, bf16, "vt")
    B, Lq, D = q.shape
    H = D // 128
    assert D == H * 128 and q.stride(2) == 1 and k.stride(2) == 1
    Lk = k.shape[1]
    Lkp = (Lk + 63) // 64 * 64
    assert vt.shape[-1] == Lkp and vt.shape[-2] == 128 and vt.shape[-3] == H, (vt.shape, Lkp, H)
    assert vt[0].is_contiguous()
    if out is None:
        out = torch.empty(B, Lq, D, device=q.device, dtype=bf16)
    _chk(out, bf16, "flash_attn.out")
    assert out.stride(2) == 1
    k_bs = 0 if (k_broadcast or k.shape[0] == 1) else k.stride(0)
    vt_bs = 0 if (k_broadcast or vt.shape[0] == 1) else vt.stride(0)
    if scale is None:
        scale = 1.0 / math.sqrt(128)
    if q_prescaled:
        scale = ATTN_Q_PRESCALED
    L.call("scail_flash_attn_bf16", q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k_seg_stride, k_bs,
           k.stride(1), vt.data_ptr(), vt_seg_stride, vt_bs, out.data_ptr(), out.stride(0), out.stride(1), B, H, Lq,
           Lk, n_seg, scale, 1 if accumulate else 0, _stream())
    return out


def cross_attn2(q, k1, vt1, k2, vt2, out=None, scale=None, q_prescaled=False):
    """softmax(q k1^T) v1 + softmax(q k2^T) v2 in one launch (text + CLIP cross attention, dit_video_crossattn_sc_xc.py:1107-1203).
    q (B, Lq, H*128) view; k1 / k2 (B|1, Lk, H*128) views; vt1 / vt2 (B|1, H, 128, ceil64(Lk)) transpose_v images.
    ``q_prescaled``: q already carries scale * log2(e) (rmsnorm_rope(..., out_scale=ATTN_LOG2_SCALE))."""
    _chk(q, bf16, "cross_attn2.q")
    B, Lq, D = q.shape
    H = D // 128
    assert D == H * 128 and q.stride(2) == 1
    if out is None:
        out = torch.empty(B, Lq, D, device=q.device, dtype=bf16)
    _chk(out, bf16, "cross_attn2.out")
    assert out.stride(2) == 1 and out.shape == (B, Lq, D)
    sets = []
    for k, vt in ((k1, vt1), (k2, vt2)):
        _chk(k, bf16, "cross_attn2.k"); _chk(vt, bf16, "cross_attn2.vt")
        Lk = k.shape[1]
        assert k.shape[0] in (1, B) and vt.shape[0] == k.shape[0] and k.shape[2] == D and k.stride(2) == 1
        assert vt.shape[1:] == (H, 128, (Lk + 63) // 64 * 64) and vt[0].is_contiguous(), (vt.shape, Lk)
        sets += [k.data_ptr(), 0 if k.shape[0] == 1 else k.stride(0), k.stride(1), vt.data_ptr(), 0 if vt.shape[0] == 1 else vt.stride(0), Lk]
    if scale is None:
        scale = 1.0 / math.sqrt(128)
    if q_prescaled:
        scale = ATTN_Q_PRESCALED
    L.call("scail_cross_attn2_bf16", q.data_ptr(), q.stride(0), q.stride(1), *sets, out.data_ptr(), out.stride(0), out.stride(1),
           B, H, Lq, scale, _stream())
    return out


def timestep_embedding(t, dim):
    _chk(t, f32, "timestep_embedding.t")
    out = torch.empty(t.shape[0], dim, device=t.device, dtype=f32)
    L.call("scail_timestep_embedding", t.data_ptr(), out.data_ptr(), t.shape[0], dim, _stream())
    return out


def small_linear(x, w, b=None, act_in=L.ACT_NONE, act_out=L.ACT_NONE):
    _chk(x, f32, "small_linear.x"); _chk(w, bf16, "small_linear.w")
    assert x.is_contiguous() and w.is_contiguous()
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=f32)
    L.call("scail_small_linear", x.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), M, N, K, act_in, act_out,
           _stream())
    return y


def adaln_table(emb, table):
    """out[l, b, :] = emb[b, :] + table[l, :]; emb (B, W) fp32, table (layers, W) fp32."""
    _chk(emb, f32, "adaln_table.emb"); _chk(table, f32, "adaln_table.table")
    assert emb.is_contiguous() and table.is_contiguous()
    out = torch.empty(table.shape[0], emb.shape[0], emb.shape[1], device=emb.device, dtype=f32)
    L.call("scail_adaln_table", emb.data_ptr(), table.data_ptr(), out.data_ptr(), table.shape[0], emb.shape[0],
           emb.shape[1], _stream())
    return out


def patchify(x, ref, pose, kpad=128, out=None):
    """x fp32 (B,T,16,H,W); ref bf16 (1|B,1,16,H,W); pose bf16 (1|B,T,16,H/2,W/2) -> (B, L, kpad) bf16."""
    _chk(x, f32, "patchify.x"); _chk(ref, bf16, "patchify.ref"); _chk(pose, bf16, "patchify.pose")
    assert x.is_contiguous() and ref.is_contiguous() and pose.is_contiguous()
    B, T, C, H, W = x.shape
    assert C == 16 and ref.shape[1:] == (1, 16, H, W) and pose.shape[1:] == (T, 16, H // 2, W // 2)
    Ltok = (1 + T) * (H // 2) * (W // 2) + T * (H // 4) * (W // 4)
    if out is None:
        out = torch.empty(B, Ltok, kpad, device=x.device, dtype=bf16)
    L.call("scail_patchify", x.data_ptr(), ref.data_ptr(), pose.data_ptr(), out.data_ptr(), B, ref.shape[0],
           pose.shape[0], T, H, W, kpad, _stream())
    return out


def unpatchify(tok, T, H, W, out=None):
    _chk(tok, bf16, "unpatchify.tok")
    assert tok.is_contiguous() and tok.shape[-1] == 64
    B = tok.shape[0]
    if out is None:
        out = torch.empty(B, T, 16, H, W, device=tok.device, dtype=f32)
    L.call("scail_unpatchify", tok.data_ptr(), out.data_ptr(), B, T, H, W, _stream())
    return out


def cfg_euler_(x, v, cfg_scale, dsigma):
    """In place: x += dsigma * (v[0] + cfg * (v[1] - v[0])).  x fp32 (1, ...), v fp32 (2, ...)."""
    _chk(x, f32, "cfg_euler.x"); _chk(v, f32, "cfg_euler.v")
    assert x.is_contiguous() and v.is_contiguous() and v.numel() == 2 * x.numel()
    L.call("scail_cfg_euler", x.data_ptr(), v.data_ptr(), x.numel(), float(cfg_scale), float(dsigma), _stream())
    return x


def to_bf16(x):
    _chk(x, f32, "to_bf16.x")
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=bf16)
    L.call("scail_f32_to_bf16", x.data_ptr(), y.data_ptr(), x.numel(), _stream())
    return y


def to_f32(x):
    _chk(x, bf16, "to_f32.x")
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=f32)
    L.call("scail_bf16_to_f32", x.data_ptr(), y.data_ptr(), x.numel(), _stream())
    return y


# ------------------------------------------------------------------------------------------------
# Wan2.1 VAE ops (channels-last activations)
# ------------------------------------------------------------------------------------------------
def prep_conv_weight(w: torch.Tensor, b: Optional[torch.Tensor], cin_pad: Optional[int] = None):
    """torch conv weight (Cout, Cin, kt, kh, kw) [or (Cout, Cin, kh, kw)] -> dict for conv3d_cl:
    w2 (Cout_pad8, Kpad) bf16 with k = ((dt*kh+dh)*kw+dw)*Cin_pad + c, zero padded; bias fp32 (Cout_pad8)."""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    Cout, Cin, kt, kh, kw = w.shape
    cin_pad = cin_pad or ((Cin + 7) // 8 * 8)
    n_pad = (Cout + 7) // 8 * 8
    wt = torch.zeros(n_pad, kt, kh, kw, cin_pad, device=w.device, dtype=torch.float32)
    wt[:Cout, ..., :Cin] = w.detach().float().permute(0, 2, 3, 4, 1)
    K = kt * kh * kw * cin_pad
    Kpad = (K + 63) // 64 * 64
    w2 = torch.zeros(n_pad, Kpad, device=w.device, dtype=bf16)
    w2[:, :K] = wt.reshape(n_pad, K).to(bf16)
    bias = torch.zeros(n_pad, device=w.device, dtype=f32)
    if b is not None:
        bias[:Cout] = b.detach().float()
    return dict(w=w2.contiguous(), b=bias, N=n_pad, Cout=Cout, Cin=cin_pad, Kpad=Kpad, k=(kt, kh, kw))


def conv3d_cl(x, wp, out_shape, stride=(1, 1, 1), pad=None, ups=False, out=None, ot_mul=1, ot_off=0, resid=None):
    """x (Ti,Hi,Wi,Cin) bf16 contiguous; wp from prep_conv_weight; out_shape = (To,Ho,Wo) covered by this call.
    pad = (pt, ph, pw): front/top/left padding (default: causal 'same': (kt-1, kh//2, kw//2)).
    Output tensor ``out`` (frames, Ho, Wo, N) -- frame index = to*ot_mul + ot_off."""
    import ctypes as C
    _chk(x, bf16, "conv3d_cl.x")
    assert x.is_contiguous() and x.dim() == 4 and x.shape[3] == wp["Cin"], (x.shape, wp["Cin"])
    Ti, Hi, Wi, Cin = x.shape
    To, Ho, Wo = out_shape
    kt, kh, kw = wp["k"]
    if pad is None:
        pad = (kt - 1, kh // 2, kw // 2)
    if out is None:
        assert ot_mul == 1 and ot_off == 0
        out = torch.empty(To, Ho, Wo, wp["N"], device=x.device, dtype=bf16)
    assert out.is_contiguous() and out.shape[1:3] == (Ho, Wo) and out.shape[3] >= wp["N"]
    assert (To - 1) * ot_mul + ot_off < out.shape[0]
    ldr = 0
    if resid is not None:
        _chk(resid, bf16, "conv3d_cl.resid")
        assert resid.is_contiguous() and resid.shape[:3] == out.shape[:3]
        ldr = resid.shape[3]
    geom = (C.c_int32 * 21)(Ti, Hi, Wi, Cin, To, Ho, Wo, kt, kh, kw, stride[0], stride[1], stride[2], pad[0], pad[1], pad[2],
                            1 if ups else 0, ot_mul, ot_off, wp["N"], wp["Kpad"])
    L.call("scail_conv3d_cl", x.data_ptr(), wp["w"].data_ptr(), wp["b"].data_ptr(), out.data_ptr(), out.shape[3],
           _ptr(resid), ldr, C.cast(geom, C.c_void_p), _stream())
    return out


def conv_norm_fusable(wp, x_channels: int) -> bool:
    """Whether conv3d_cl_norm covers this convolution (3x3x3, Cin % 32 == 0, at most 96 output channels)."""
    return tuple(wp["k"]) == (3, 3, 3) and x_channels % 32 == 0 and wp["N"] <= 96


def conv_norm_generated(wp, x_shape) -> bool:
    """Whether conv3d_cl_norm of an activation of shape (T, H, W, Cin) runs the generated kernel's norm epilogue (scail_conv4f_e4: 96 output
    channels; scail_conv3d_kernel_for(..., fused_norm = 1) == 4)."""
    import ctypes as C
    T, H, W, Cin = x_shape
    kt, kh, kw = wp["k"]
    geom = (C.c_int32 * 21)(T, H, W, Cin, T, H, W, kt, kh, kw, 1, 1, 1, kt - 1, kh // 2, kw // 2, 0, 1, 0, wp["N"], wp["Kpad"])
    return L.load().scail_conv3d_kernel_for(C.cast(geom, C.c_void_p), wp["N"], 0, 1) == 4


def conv_generated(wp, x_shape) -> bool:
    """Whether the causal 'same' conv3d_cl of an activation of shape (T, H, W, Cin) runs the generated kernels (scail_conv3d_kernel_for == 4).
    The VAE then prefers conv3d_cl + rms_silu over the fused conv3d_cl_norm of the hipcc halo kernel: 13.9 + 2.8 ms against 21.0 ms on the
    96-channel full-resolution shape (profiles/r03_vae_kernel_stats.md)."""
    import ctypes as C
    T, H, W, Cin = x_shape
    kt, kh, kw = wp["k"]
    geom = (C.c_int32 * 21)(T, H, W, Cin, T, H, W, kt, kh, kw, 1, 1, 1, kt - 1, kh // 2, kw // 2, 0, 1, 0, wp["N"], wp["Kpad"])
    return L.load().scail_conv3d_kernel_for(C.cast(geom, C.c_void_p), wp["N"], 0, 0) == 4


def conv3d_cl_norm(x, wp, gamma, out=None):
    """SiLU(RMS_norm(causal 3x3x3 conv(x)) * gamma) in one kernel (scail_conv3d_cl_norm): x (T,H,W,Cin) bf16 -> (T,H,W,N)."""
    import ctypes as C
    _chk(x, bf16, "conv3d_cl_norm.x"); _chk(gamma, f32, "conv3d_cl_norm.gamma")
    assert x.is_contiguous() and x.dim() == 4 and x.shape[3] == wp["Cin"] and gamma.numel() == wp["N"]
    T, H, W, Cin = x.shape
    if out is None:
        out = torch.empty(T, H, W, wp["N"], device=x.device, dtype=bf16)
    assert out.is_contiguous() and out.shape == (T, H, W, wp["N"])
    geom = (C.c_int32 * 21)(T, H, W, Cin, T, H, W, 3, 3, 3, 1, 1, 1, 2, 1, 1, 0, 1, 0, wp["N"], wp["Kpad"])
    L.call("scail_conv3d_cl_norm", x.data_ptr(), wp["w"].data_ptr(), wp["b"].data_ptr(), out.data_ptr(), out.shape[3],
           gamma.data_ptr(), C.cast(geom, C.c_void_p), _stream())
    return out


def _next_norm_geom(wp, x_shape, out_shape, pad, ups):
    import ctypes as C
    Ti, Hi, Wi, Cin = x_shape
    To, Ho, Wo = out_shape
    kt, kh, kw = wp["k"]
    if pad is None:
        pad = (kt - 1, kh // 2, kw // 2)
    return (C.c_int32 * 21)(Ti, Hi, Wi, Cin, To, Ho, Wo, kt, kh, kw, 1, 1, 1, pad[0], pad[1], pad[2], 1 if ups else 0, 1, 0, wp["N"], wp["Kpad"])


def conv_resid_norm_generated(wp, x_shape, out_shape=None, pad=None, ups=False, resid=True) -> bool:
    """Whether conv3d_cl_resid_norm of an activation of shape (T, H, W, Cin) runs ONE generated kernel (scail_conv4c_e5 / e6 with a residual,
    scail_conv4u_e7 without; scail_conv3d_kernel_for(..., fused_norm = 2) == 4) -- the rule csrc/vae_exec.hip and the layer path of wan_vae.py share."""
    import ctypes as C
    geom = _next_norm_geom(wp, x_shape, out_shape or tuple(x_shape[:3]), pad, ups)
    return L.load().scail_conv3d_kernel_for(C.cast(geom, C.c_void_p), wp["N"], wp["N"] if resid else 0, 2) != 0       # 4: generated kernel, 2: direct-gather dual form


def conv3d_cl_resid_norm(x, wp, resid, gamma, want_raw=True, out_shape=None, pad=None, ups=False):
    """(resid + conv(x), SiLU(RMS_norm(that) * gamma)) in one call (scail_conv3d_cl_resid_norm): the last convolution of a ResidualBlock -- or, with
    resid = None, any stride-1 convolution such as Resample's behind the 2x upsample -- with the next consumer's norm.  Returns
    (raw | None, normalised), both (To, Ho, Wo, N) bf16."""
    import ctypes as C
    _chk(x, bf16, "conv3d_cl_resid_norm.x"); _chk(gamma, f32, "conv3d_cl_resid_norm.gamma")
    assert x.is_contiguous() and x.dim() == 4 and x.shape[3] == wp["Cin"] and gamma.numel() == wp["N"]
    To, Ho, Wo = out_shape or tuple(x.shape[:3])
    if resid is not None:
        _chk(resid, bf16, "conv3d_cl_resid_norm.resid")
        assert resid.is_contiguous() and resid.shape == (To, Ho, Wo, wp["N"])
    else:
        assert want_raw, "without a residual the raw output is part of the contract"
    raw = torch.empty(To, Ho, Wo, wp["N"], device=x.device, dtype=bf16) if want_raw else None
    nrm = torch.empty(To, Ho, Wo, wp["N"], device=x.device, dtype=bf16)
    geom = _next_norm_geom(wp, x.shape, (To, Ho, Wo), pad, ups)
    L.call("scail_conv3d_cl_resid_norm", x.data_ptr(), wp["w"].data_ptr(), wp["b"].data_ptr(), _ptr(raw), nrm.data_ptr(), wp["N"],
           _ptr(resid), resid.shape[3] if resid is not None else 0, gamma.data_ptr(), C.cast(geom, C.c_void_p), _stream())
    return raw, nrm


def rms_silu(x, gamma, silu=True, out=None):
    """x (..., C) channels-last bf16 contiguous; gamma fp32 (C)."""
    _chk(x, bf16, "rms_silu.x"); _chk(gamma, f32, "rms_silu.gamma")
    assert x.is_contiguous()
    Cc = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    L.call("scail_rms_silu", x.data_ptr(), out.data_ptr(), gamma.data_ptr(), x.numel() // Cc, Cc, 1 if silu else 0, _stream())
    return out


def softmax_rows_(s, n, scale):
    """In place softmax(scale * s[:, :n]) over rows of a 2-D bf16 tensor (row stride s.stride(0))."""
    _chk(s, bf16, "softmax_rows.s")
    assert s.dim() == 2 and s.stride(1) == 1
    L.call("scail_softmax_rows", s.data_ptr(), s.stride(0), s.shape[0], n, float(scale), _stream())
    return s


def transpose2d(x, out):
    """x (B, R, C) view (last dim contiguous) -> out (B, C, >=R) (writes the first R columns)."""
    _chk(x, bf16, "transpose2d.x"); _chk(out, bf16, "transpose2d.out")
    B, R, Cc = x.shape
    assert x.stride(2) == 1 and out.stride(2) == 1 and out.shape[0] == B and out.shape[1] == Cc and out.shape[2] >= R
    L.call("scail_transpose2d", x.data_ptr(), x.stride(1), x.stride(0), out.data_ptr(), out.stride(1), out.stride(0), R, Cc, B,
           _stream())
    return out


def to_channels_last(x, cpad, a=None, b=None):
    """x fp32 (C, T, H, W) -> (T, H, W, cpad) bf16, y = x*a[c] + b[c]."""
    _chk(x, f32, "to_channels_last.x")
    x = x.contiguous()
    Cc, T, H, W = x.shape
    y = torch.empty(T, H, W, cpad, device=x.device, dtype=bf16)
    L.call("scail_to_channels_last", x.data_ptr(), y.data_ptr(), _ptr(a), _ptr(b), Cc, cpad, T * H * W, _stream())
    return y


def from_channels_last(x, Cc, a=None, b=None, lo=-3.0e38, hi=3.0e38):
    """x bf16 (T, H, W, ld) -> fp32 (Cc, T, H, W): clamp((x + b[c]) * a[c], lo, hi)."""
    _chk(x, bf16, "from_channels_last.x")
    assert x.is_contiguous()
    T, H, W, ld = x.shape
    y = torch.empty(Cc, T, H, W, device=x.device, dtype=f32)
    L.call("scail_from_channels_last", x.data_ptr(), ld, y.data_ptr(), _ptr(a), _ptr(b), Cc, T * H * W, float(lo), float(hi),
           _stream())
    return y


# ------------------------------------------------------------------------------------------------
# conditioning-encoder ops
# ------------------------------------------------------------------------------------------------
def attn_small(q, k, v, heads, scale=1.0, bucket=None, bias_tab=None, key_mask=None, out=None):
    """q (B, Lq, H*hd), k/v (B, Lk, H*hd) bf16 views (last dim contiguous) -> (B, Lq, H*hd).
    bucket int32 (Lq, Lk) + bias_tab fp32 (n_buckets, H); key_mask int32 (B, Lk), 0 = excluded."""
    import ctypes as C
    _chk(q, bf16, "attn_small.q"); _chk(k, bf16, "attn_small.k"); _chk(v, bf16, "attn_small.v")
    B, Lq, D = q.shape
    Lk = k.shape[1]
    hd = D // heads
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        out = torch.empty(B, Lq, D, device=q.device, dtype=bf16)
    st = (C.c_int64 * 8)(q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1))
    if bucket is not None:
        assert bucket.dtype == torch.int32 and bucket.is_contiguous() and bucket.shape == (Lq, Lk)
        _chk(bias_tab, f32, "attn_small.bias_tab")
        assert bias_tab.is_contiguous() and bias_tab.shape[1] == heads
    if key_mask is not None:
        assert key_mask.dtype == torch.int32 and key_mask.shape == (B, Lk) and key_mask.stride(1) == 1
    L.call("scail_attn_small", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), C.cast(st, C.c_void_p), B, heads, Lq, Lk,
           hd, float(scale), _ptr(bucket), _ptr(bias_tab), _ptr(key_mask), key_mask.stride(0) if key_mask is not None else 0,
           _stream())
    return out


def mul_(a, b, out=None):
    _chk(a, bf16, "mul.a"); _chk(b, bf16, "mul.b")
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    out = a if out is None else out
    L.call("scail_mul_bf16", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
    return out


def row_affine(x, rowscale=None, addrow=None, out=None):
    """x (rows.., D) bf16 contiguous: y[r] = x[r] * rowscale[r] + addrow[r % addrow.rows]."""
    _chk(x, bf16, "row_affine.x")
    assert x.is_contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    out = torch.empty_like(x) if out is None else out
    ar = 0
    if addrow is not None:
        _chk(addrow, bf16, "row_affine.addrow")
        assert addrow.is_contiguous() and addrow.shape[-1] == D
        ar = addrow.numel() // D
    if rowscale is not None:
        _chk(rowscale, f32, "row_affine.rowscale")
        assert rowscale.numel() == rows and rowscale.is_contiguous()
    L.call("scail_row_affine", x.data_ptr(), out.data_ptr(), _ptr(rowscale), _ptr(addrow), ar, rows, D, _stream())
    return out

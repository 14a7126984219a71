"""Binding of the C-level VAE executor (include/scail_vae.h, csrc/vae_exec.hip): ``WanVAE_.encode`` / ``.decode`` as ONE
call into libscail_hip.so each.  The layer-by-layer orchestration in ``wan_vae.py`` stays as the cross-check (same kernels,
same order: bit-identical results)."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import lib as L

_p, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64


class ConvW(C.Structure):
    _fields_ = [("w", _p), ("b", _p), ("Cin", _i32), ("N", _i32), ("Kpad", _i32), ("kt", _i32), ("kh", _i32), ("kw", _i32)]


class Res(C.Structure):
    _fields_ = [("gamma0", _p), ("conv2", ConvW), ("gamma3", _p), ("conv6", ConvW), ("shortcut", ConvW)]


class Attn(C.Structure):
    _fields_ = [("gamma", _p), ("q_w", _p), ("k_w", _p), ("v_w", _p), ("proj_w", _p),
                ("q_b", _p), ("k_b", _p), ("v_b", _p), ("proj_b", _p), ("C", _i32)]


class Stage(C.Structure):
    _fields_ = [("kind", _i32), ("temporal", _i32), ("res", Res), ("resample", ConvW), ("time_conv", ConvW),
                ("time_conv0", ConvW), ("time_conv1", ConvW)]


class Weights(C.Structure):
    _fields_ = [("z_dim", _i32),
                ("enc_conv1", ConvW), ("enc", C.POINTER(Stage)), ("n_enc", _i32),
                ("enc_mid0", Res), ("enc_attn", Attn), ("enc_mid2", Res), ("enc_head_gamma", _p), ("enc_head", ConvW),
                ("conv1", ConvW), ("enc_scale", _p), ("enc_shift", _p),
                ("dec_scale", _p), ("dec_shift", _p), ("conv2", ConvW), ("dec_conv1", ConvW),
                ("dec_mid0", Res), ("dec_attn", Attn), ("dec_mid2", Res), ("dec", C.POINTER(Stage)), ("n_dec", _i32),
                ("dec_head_gamma", _p), ("dec_head", ConvW)]


def _conv(wp) -> ConvW:
    if wp is None:
        return ConvW(None, None, 0, 0, 0, 0, 0, 0)
    kt, kh, kw = wp["k"]
    return ConvW(wp["w"].data_ptr(), wp["b"].data_ptr(), wp["Cin"], wp["N"], wp["Kpad"], kt, kh, kw)


def _res(W: Dict, n: str) -> Res:
    return Res(W[n + ".residual.0.gamma"].data_ptr(), _conv(W[n + ".residual.2"]), W[n + ".residual.3.gamma"].data_ptr(),
               _conv(W[n + ".residual.6"]), _conv(W.get(n + ".shortcut")))


def _attn(W: Dict, n: str) -> Attn:
    (qw, qb), (kw, kb), (vw, vb), (pw, pb) = W[n + ".q"], W[n + ".k"], W[n + ".v"], W[n + ".proj"]
    return Attn(W[n + ".norm.gamma"].data_ptr(), qw.data_ptr(), kw.data_ptr(), vw.data_ptr(), pw.data_ptr(),
                qb.data_ptr(), kb.data_ptr(), vb.data_ptr(), pb.data_ptr(), qw.shape[0])


_TRACE_FN = C.CFUNCTYPE(None, _p, C.c_int, C.c_char_p, _p, _i64, _i64, _i64, _i64)


class CVae:
    def __init__(self, model, W: Dict):
        L.load()
        self._keep = [W]
        dev = W["mean"].device
        enc_plan, dec_plan = model.encoder_plan(), model.decoder_plan()
        enc = (Stage * max(len(enc_plan), 1))()
        for i, (kind, n, a, b) in enumerate(enc_plan):
            if kind == "res":
                enc[i].kind, enc[i].res = 0, _res(W, n)
            else:
                enc[i].kind, enc[i].temporal, enc[i].resample = 1, int(bool(b)), _conv(W[n + ".resample.1"])
                if b:
                    enc[i].time_conv = _conv(W[n + ".time_conv"])
        dec = (Stage * max(len(dec_plan), 1))()
        for i, (kind, n, a, b) in enumerate(dec_plan):
            if kind == "res":
                dec[i].kind, dec[i].res = 0, _res(W, n)
            else:
                dec[i].kind, dec[i].temporal, dec[i].resample = 2, int(bool(b)), _conv(W[n + ".resample.1"])
                if b:
                    dec[i].time_conv0, dec[i].time_conv1 = _conv(W[n + ".time_conv#0"]), _conv(W[n + ".time_conv#1"])
        inv_std, neg_mean = (1.0 / W["std"]).contiguous(), (-W["mean"]).contiguous()
        self._keep += [enc, dec, inv_std, neg_mean]
        w = Weights()
        w.z_dim = model.z_dim
        w.enc_conv1, w.enc, w.n_enc = _conv(W["encoder.conv1"]), enc, len(enc_plan)
        w.enc_mid0, w.enc_attn, w.enc_mid2 = _res(W, "encoder.middle.0"), _attn(W, "encoder.middle.1"), _res(W, "encoder.middle.2")
        w.enc_head_gamma, w.enc_head, w.conv1 = W["encoder.head.0.gamma"].data_ptr(), _conv(W["encoder.head.2"]), _conv(W["conv1"])
        w.enc_scale, w.enc_shift = inv_std.data_ptr(), neg_mean.data_ptr()
        w.dec_scale, w.dec_shift = W["std"].data_ptr(), W["mean"].data_ptr()
        w.conv2, w.dec_conv1 = _conv(W["conv2"]), _conv(W["decoder.conv1"])
        w.dec_mid0, w.dec_attn, w.dec_mid2 = _res(W, "decoder.middle.0"), _attn(W, "decoder.middle.1"), _res(W, "decoder.middle.2")
        w.dec, w.n_dec = dec, len(dec_plan)
        w.dec_head_gamma, w.dec_head = W["decoder.head.0.gamma"].data_ptr(), _conv(W["decoder.head.2"])
        h = _p()
        L.call("scail_vae_create", C.byref(w), C.byref(h))
        self._h, self._ws, self._dev, self.z = h, None, dev, model.z_dim

    def set_trace(self, fn):
        """include/scail_vae.h ``scail_vae_set_trace``: ``fn(index, op, tensor)`` after every launch that completes an activation; ``tensor`` is
        a (T, H, W, C) bf16 VIEW into the workspace, ordered on torch's current stream (the one encode / decode enqueue on): read or copy
        it with torch ops inside the callback, do not keep it.  ``None`` switches tracing off."""
        if fn is None:
            self._trace_c = None
            L.call("scail_vae_set_trace", self._h, None, None)
            return

        def thunk(_user, index, op, data, T, H, W, Cc):
            off = data - self._ws.data_ptr()
            view = self._ws[off:off + T * H * W * Cc * 2].view(torch.bfloat16).view(T, H, W, Cc)
            fn(index, op.decode(), view)

        self._trace_c = _TRACE_FN(thunk)
        L.call("scail_vae_set_trace", self._h, C.cast(self._trace_c, _p), None)

    def close(self):
        if self._h is not None:
            L.load().scail_vae_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _workspace(self, T, H, W):
        need = L.load().scail_vae_workspace_bytes(self._h, T, H, W)
        if need < 0:
            raise ValueError("video needs T = 1 + 4n frames and H, W multiples of 8")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, device=self._dev, dtype=torch.uint8)
        return self._ws

    def encode(self, video: torch.Tensor) -> torch.Tensor:
        """video fp32 (3, T, H, W) on the GPU -> (z, 1 + (T-1)/4, H/8, W/8) fp32."""
        _, T, H, W = video.shape
        ws = self._workspace(T, H, W)
        out = torch.empty(self.z, 1 + (T - 1) // 4, H // 8, W // 8, device=video.device, dtype=torch.float32)
        L.call("scail_vae_encode", self._h, video.data_ptr(), out.data_ptr(), T, H, W, ws.data_ptr(), ws.numel(),
               torch.cuda.current_stream().cuda_stream)
        return out

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z fp32 (zc, Tl, h, w) -> video fp32 (3, 1 + 4 (Tl - 1), 8h, 8w), not clamped."""
        _, Tl, h, w = z.shape
        T, H, W = 1 + 4 * (Tl - 1), 8 * h, 8 * w
        ws = self._workspace(T, H, W)
        out = torch.empty(3, T, H, W, device=z.device, dtype=torch.float32)
        L.call("scail_vae_decode", self._h, z.data_ptr(), out.data_ptr(), Tl, h, w, ws.data_ptr(), ws.numel(),
               torch.cuda.current_stream().cuda_stream)
        return out

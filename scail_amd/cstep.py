"""Binding of the C-level step executor (include/scail_dit.h, csrc/dit_step.hip): the whole network evaluation of one
sampler step is ONE call into libscail_hip.so -- the host only hands over device pointers.  Used by
``DiffusionTransformer`` when ``use_c_step`` is set (single sequence-parallel rank); the Python orchestration in
``dit._run`` stays as the multi-rank / instrumented path and as the cross-check (both enqueue the same kernels in the same
order, so their results are bit-identical)."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import lib as L

_p, _i64 = C.c_void_p, C.c_int64


class DitConfig(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("num_heads", C.c_int32), ("inner_hidden_size", C.c_int32),
                ("num_layers", C.c_int32), ("text_dim", C.c_int32), ("clip_dim", C.c_int32),
                ("time_freq_dim", C.c_int32), ("time_embed_dim", C.c_int32), ("layernorm_epsilon", C.c_float)]


_LAYER_FIELDS = ["qkv_w", "qkv_b", "o_w", "o_b", "qn", "kn", "cq_w", "cq_b", "co_w", "co_b", "cqn", "ln_w", "ln_b",
                 "w1", "b1", "w2", "b2"]


class DitLayer(C.Structure):
    _fields_ = [(n, _p) for n in _LAYER_FIELDS]


class DitWeights(C.Structure):
    _fields_ = [(n, _p) for n in ("patch_w", "patch_b", "pose_w", "pose_b", "time0_w", "time0_b", "time2_w", "time2_b",
                                  "adaln_w", "adaln_b", "adaln_tables", "final_table", "final_w", "final_b")] + \
               [("layers", C.POINTER(DitLayer))]


class DitCond(C.Structure):
    _fields_ = [("k_text", _p), ("vt_text", _p), ("k_clip", _p), ("vt_clip", _p), ("Lt", _i64), ("Lc", _i64), ("Bc", _i64)]


# include/scail_dit.h "sequence-parallel execution": the exchange callback and its descriptor
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p)
SP_ALLGATHER, SP_ULYSSES = 0, 1
SP_FWD_START, SP_FWD_WAIT, SP_BACK_START, SP_BACK_WAIT = 0, 1, 2, 3


class DitSp(C.Structure):
    _fields_ = [("ranks", C.c_int32), ("mode", C.c_int32), ("send", _p), ("recv", _p), ("ofull", _p), ("back", _p),
                ("exchange", EXCHANGE_FN), ("user", _p), ("side_stream", _p * 2)]


def _cond_struct(cond: Dict) -> "DitCond":
    k_text, k_clip = cond["k_text"], cond["k_clip"]
    return DitCond(k_text.data_ptr(), cond["vt_text"].data_ptr(), k_clip.data_ptr(), cond["vt_clip"].data_ptr(),
                   k_text.shape[2], k_clip.shape[2], k_clip.shape[1])


class CStep:
    """Handle around scail_dit_create / scail_dit_step for one prepared network (``net.prepare()`` dict)."""

    def __init__(self, net, W: Dict):
        L.load()
        self._keep = W                                   # the pointer tables reference these tensors
        cfg = DitConfig(net.hidden_size, net.num_attention_heads, net.inner_hidden_size, net.num_layers, net.text_dim,
                        1280, net.time_freq_dim, net.time_embed_dim, float(net.layernorm_epsilon))
        layers = (DitLayer * net.num_layers)()
        for i, lw in enumerate(W["layers"]):
            for n in _LAYER_FIELDS:
                setattr(layers[i], n, lw[n].data_ptr())
        w = DitWeights(W["patch_w"].data_ptr(), W["patch_b"].data_ptr(), W["pose_w"].data_ptr(), W["pose_b"].data_ptr(),
                       W["time_embed.0.w"].data_ptr(), W["time_embed.0.b"].data_ptr(),
                       W["time_embed.2.w"].data_ptr(), W["time_embed.2.b"].data_ptr(),
                       W["adaln_projection.1.w"].data_ptr(), W["adaln_projection.1.b"].data_ptr(),
                       W["adaln_tables"].data_ptr(), W["final_table"].data_ptr(), W["final_w"].data_ptr(),
                       W["final_b"].data_ptr(), layers)
        h = _p()
        L.call("scail_dit_create", C.byref(cfg), C.byref(w), C.byref(h))
        self._h = h
        self._ws = None

    def close(self):
        if self._h is not None:
            L.load().scail_dit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    PROF_SELF_ATTN, PROF_GEMM, PROF_CROSS_ATTN = 0, 1, 2          # include/scail_dit.h SCAIL_DIT_PROF_*
    PROF_XCH_FWD_WAIT, PROF_XCH_BACK_WAIT = 3, 4                  # exposed part of the sequence-parallel exchange (stream waits)
    PROF_ATTN_RESTARTS = 5                                        # not a time: restarted self-attention workgroups (second value of profile_read)

    def profile(self, enable: bool) -> None:
        """HIP-event timing of the executor's own launches (scail_dit_profile): on = restart the counters."""
        L.call("scail_dit_profile", self._h, 1 if enable else 0)

    def profile_read(self, category: int):
        """(summed kernel ms, launches) of one category since profile(True); waits for the recorded events."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        L.call("scail_dit_profile_read", self._h, category, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def workspace_bytes(self, B, T, H, W) -> int:
        n = L.load().scail_dit_workspace_bytes(self._h, B, T, H, W)
        if n < 0:
            raise L.ScailHipError("scail_dit_workspace_bytes: bad shape")
        return n

    CFG_PAIR = 1                                                  # include/scail_dit.h SCAIL_DIT_CFG_PAIR

    def step(self, x32, t32, cond: Dict, ref, pose, cos, sin, cfg_pair: bool = False) -> torch.Tensor:
        """``cfg_pair``: the caller states that x32 / t32 hold the same latent and timestep twice (VanillaCFG's batch of 2): layer 0 up to
        its first cross attention is evaluated once (SCAIL_DIT_CFG_PAIR; bit-identical on such inputs)."""
        B, T, _, H, W = x32.shape
        need = self.workspace_bytes(B, T, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x32.device:
            self._ws = torch.empty(need, device=x32.device, dtype=torch.uint8)
        k_text, k_clip = cond["k_text"], cond["k_clip"]
        cc = DitCond(k_text.data_ptr(), cond["vt_text"].data_ptr(), k_clip.data_ptr(), cond["vt_clip"].data_ptr(),
                     k_text.shape[2], k_clip.shape[2], k_clip.shape[1])
        out = torch.empty(B, T, 16, H, W, device=x32.device, dtype=torch.float32)
        L.call("scail_dit_step", self._h, x32.data_ptr(), t32.data_ptr(), C.byref(cc), ref.data_ptr(), ref.shape[0],
               pose.data_ptr(), pose.shape[0], cos.data_ptr(), sin.data_ptr(), out.data_ptr(), B, T, H, W,
               self.CFG_PAIR if cfg_pair else 0, self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream().cuda_stream)
        return out

    def sample(self, x32, sigmas, cfg_scale, cond: Dict, ref, pose, cos, sin) -> torch.Tensor:
        """The whole Euler loop in one C call (scail_dit_sample).  x32 (1,T,16,H,W) fp32 is updated in place and returned;
        ``sigmas`` is the host schedule (n_steps + 1 values); ``cond`` the batch-2 conditioning (uncond, cond)."""
        import ctypes as C2
        _, T, _, H, W = x32.shape
        lib = L.load()
        need = lib.scail_dit_sample_workspace_bytes(self._h, T, H, W)
        if need < 0:
            raise L.ScailHipError("scail_dit_sample_workspace_bytes: bad shape")
        if self._ws is None or self._ws.numel() < need or self._ws.device != x32.device:
            self._ws = torch.empty(need, device=x32.device, dtype=torch.uint8)
        sig = sigmas.float().cpu()
        n = sig.numel() - 1
        ts = (sig[:-1] * 1000.0).repeat_interleave(2).to(x32.device).contiguous()       # (n, 2) device fp32
        ds = (sig[1:] - sig[:-1]).contiguous()
        dsa = (C2.c_float * n)(*[float(v) for v in ds])
        k_text, k_clip = cond["k_text"], cond["k_clip"]
        cc = DitCond(k_text.data_ptr(), cond["vt_text"].data_ptr(), k_clip.data_ptr(), cond["vt_clip"].data_ptr(),
                     k_text.shape[2], k_clip.shape[2], k_clip.shape[1])
        assert x32.is_contiguous() and x32.dtype == torch.float32 and ref.shape[0] == 1 and pose.shape[0] == 1
        L.call("scail_dit_sample", self._h, x32.data_ptr(), ts.data_ptr(), C2.cast(dsa, C2.c_void_p), n, float(cfg_scale),
               C2.byref(cc), ref.data_ptr(), pose.data_ptr(), cos.data_ptr(), sin.data_ptr(), T, H, W,
               self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream().cuda_stream)
        return x32

    def block(self, layer: int, hidden: torch.Tensor, mod: torch.Tensor, cond: Dict, cos, sin) -> torch.Tensor:
        """Seam B2: one transformer block in place on ``hidden`` (B, Ltok, D) bf16; ``mod`` (B, 6D) fp32 = adaLN embedding +
        this layer's table.  Returns ``hidden``."""
        B, Ltok, _ = hidden.shape
        lib = L.load()
        need = lib.scail_dit_block_workspace_bytes(self._h, B, Ltok)
        if self._ws is None or self._ws.numel() < need or self._ws.device != hidden.device:
            self._ws = torch.empty(need, device=hidden.device, dtype=torch.uint8)
        k_text, k_clip = cond["k_text"], cond["k_clip"]
        cc = DitCond(k_text.data_ptr(), cond["vt_text"].data_ptr(), k_clip.data_ptr(), cond["vt_clip"].data_ptr(),
                     k_text.shape[2], k_clip.shape[2], k_clip.shape[1])
        assert hidden.is_contiguous() and mod.is_contiguous() and mod.dtype == torch.float32
        L.call("scail_dit_block", self._h, layer, hidden.data_ptr(), mod.data_ptr(), C.byref(cc), cos.data_ptr(), sin.data_ptr(),
               B, Ltok, self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream().cuda_stream)
        return hidden

    # ---- sequence-parallel rank (include/scail_dit.h: scail_dit_step_sp / scail_dit_block_sp) ----
    def _sp_call(self, name, xch, *args):
        """Run one executor call whose collectives go through ``xch`` (scail_amd.parallel.CExchange); an exception raised inside the
        callback is re-raised here (the executor only sees a non-zero status)."""
        xch.error = None
        try:
            L.call(name, *args)
        except L.ScailHipError:
            if xch.error is not None:
                raise xch.error
            raise

    def _ws_for(self, need, device):
        if need < 0:
            raise L.ScailHipError("sequence-parallel workspace: bad shape / mode")
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, device=device, dtype=torch.uint8)
        return self._ws

    def step_sp(self, x32, t32, cond: Dict, ref, pose, cos, sin, xch, cfg_pair: bool = False) -> torch.Tensor:
        """One network evaluation on this rank's latent slab (scail_dit_step_sp); ``xch`` owns the exchange buffers and issues the
        collectives from the executor's callback."""
        B, T, _, H, W = x32.shape
        ws = self._ws_for(L.load().scail_dit_sp_workspace_bytes(self._h, xch.mode_code, xch.size, B, T, H, W), x32.device)
        cc, sp = _cond_struct(cond), xch.descriptor()
        out = torch.empty(B, T, 16, H, W, device=x32.device, dtype=torch.float32)
        self._sp_call("scail_dit_step_sp", xch, self._h, x32.data_ptr(), t32.data_ptr(), C.byref(cc), ref.data_ptr(), ref.shape[0],
                      pose.data_ptr(), pose.shape[0], cos.data_ptr(), sin.data_ptr(), out.data_ptr(), B, T, H, W, C.byref(sp),
                      self.CFG_PAIR if cfg_pair else 0, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        return out

    def block_sp(self, layer: int, hidden: torch.Tensor, mod: torch.Tensor, cond: Dict, cos, sin, xch) -> torch.Tensor:
        """Seam B2 for a sequence-parallel rank: one block in place on this rank's ``hidden`` (B, Ltok, D)."""
        B, Ltok, _ = hidden.shape
        ws = self._ws_for(L.load().scail_dit_block_sp_workspace_bytes(self._h, xch.mode_code, xch.size, B, Ltok), hidden.device)
        cc, sp = _cond_struct(cond), xch.descriptor()
        assert hidden.is_contiguous() and mod.is_contiguous() and mod.dtype == torch.float32
        self._sp_call("scail_dit_block_sp", xch, self._h, layer, hidden.data_ptr(), mod.data_ptr(), C.byref(cc), cos.data_ptr(), sin.data_ptr(),
                      B, Ltok, C.byref(sp), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        return hidden

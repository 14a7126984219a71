"""SAT mixins: the HIP hot path hooked into the REFERENCE's own network object (seams B2 / B3 of SURVEY.md 8b).

The reference composes its DiT from hooks: ``BaseModel.add_mixin`` / ``collect_hooks_`` (sat/model/base_model.py:114-176)
gather the methods named in ``HOOKS_DEFAULT`` (sat/transformer_defaults.py:296-306) from the registered mixins, and the
SAT transformer calls ``hooks['layer_forward']`` per layer (sat/model/transformer.py:712-719) and ``hooks['attention_fn']``
inside ``attention_forward`` / ``cross_attention_forward`` (dit_video_crossattn_sc_xc.py:1058-1105, 1107-1203).  A
maintainer who wants to keep the reference's ``DiffusionTransformer`` object (checkpoint loading, training code, other
mixins) and only swap the hot path adds one of these mixins AFTER the model is built:

    from scail_amd import sat_mixins
    sat_mixins.install(model.model.diffusion_model, seam="block")        # or seam="attention"

  * ``HipLayerMixin.layer_forward``   (seam B2) replaces ``AdaLNMixin.layer_forward`` (dit...:1009-1051): one call of
    ``scail_dit_block`` (include/scail_dit.h) per layer on the reference's hidden states, with the reference's own
    parameters (shared storage, no copy), its AdaLN embedding ``emb``, its embedded text / CLIP conditioning
    (``encoder_outputs`` / ``image_clip_features``) and the RoPE window its forward selects (``rope_*`` kwargs).
  * ``HipAttentionMixin.attention_fn`` (seam B3) replaces ``attention_fn_default`` / ``UlyessAttentionMixin.attention_fn``
    (sat/transformer_defaults.py:47-79, dit...:351-379): q, k, v arrive as (B, heads, L, 128) after the rotary hook and the
    context goes back in the same layout; everything else of the block stays the reference's torch code.

Both hooks are marked ``non_conflict`` -- SAT's sanctioned way to stack a hook on one that is already registered (the DiT
registers ``ulysse``, ``pos_embed`` and ``adaln_layer`` in its constructor, dit...:1384-1418).  ``layer_forward`` takes
``old_impl`` and does not call it.  ``attention_fn`` is a CHAIN in this network: ``pos_embed`` contributes a non_conflict
``attention_fn`` that applies the 3-segment RoPE to q / k and then calls its ``old_impl`` (dit...:653-757), the innermost
link being ``UlyessAttentionMixin.attention_fn``.  A mixin added later lands OUTSIDE that chain, so the HIP hook re-links
it: every non_conflict wrapper stays, only the innermost link (the SDPA / Ulysses call) is replaced by the HIP kernel.
Unsupported situations (CPU tensors, dropout in training, masks, head_dim != 128, a SAT sequence-parallel world > 1 in
EITHER seam -- both replace code that exchanges between ranks) raise ``ScailHipError``: there is no fallback to the torch path.

The compute itself sits behind a small backend object (``HipBackend``: the ctypes bindings of scail_amd.ops / cstep) so a
CPU test can hook the mixins into the real reference network and check the hook table and the argument contract with a
stand-in backend (tests/test_sat_mixins_cpu.py), and a GPU test feeds the same tensors through ``HipBackend``
(tests/test_sat_mixins_gpu.py).
"""
from __future__ import annotations

import sys
from typing import Dict, Optional

import torch
from torch import nn

from . import lib as L


# ------------------------------------------------------------------------------------------------
# SAT's BaseMixin when the host program has SAT loaded (it must have, to own a model to hook); otherwise a stand-in with
# the same surface so the hook functions stay callable (GPU box: no SAT).  scail_amd never imports SAT itself.
# ------------------------------------------------------------------------------------------------
def _sat_bases():
    m = sys.modules.get("sat.model.base_model")
    if m is not None and hasattr(m, "BaseMixin"):
        return m.BaseMixin, m.non_conflict

    class BaseMixin(nn.Module):                       # sat/model/base_model.py:48-78
        def reinit(self, parent_model=None):
            pass

    def non_conflict(func):                           # sat/model/base_model.py:32-38
        func.non_conflict = True
        return func

    return BaseMixin, non_conflict


def _sequence_parallel_world_size() -> int:
    """SAT's sequence-parallel world size when the host program has SAT's mpu loaded and initialised (sat/mpu/initialize.py), else
    1.  Queried through sys.modules: scail_amd never imports SAT itself."""
    for name in ("sat.mpu", "sat.mpu.initialize"):
        m = sys.modules.get(name)
        f = getattr(m, "get_sequence_parallel_world_size", None) if m is not None else None
        if f is not None:
            try:
                return int(f())
            except Exception:            # mpu not initialised (single process): SAT itself asserts here
                return 1
    return 1


def _require_single_sp_rank(where: str):
    """Both seams replace code that EXCHANGES under sequence parallelism (the Ulysses all-to-alls of
    UlyessAttentionMixin.attention_fn, dit...:351-379 / sat/mpu/ulysses_attn_layer.py:41-110): with world > 1 each rank would
    silently attend to its own token shard only.  The multi-rank path of this repo is scail_amd.parallel.SequenceParallel behind
    the network seam (B1), not the hook seams."""
    n = _sequence_parallel_world_size()
    if n > 1:
        raise L.ScailHipError(f"{where}: sequence-parallel world size is {n}; the SAT hook seams (B2 / B3) are single-rank -- use the "
                              f"network seam (network_config.target: scail_amd.dit.DiffusionTransformer) with "
                              f"scail_amd.parallel.SequenceParallel for sequence parallelism")


def _splice_innermost(chain, inner):
    """collect_hooks_ (sat/model/base_model.py:151-163) nests non_conflict hooks as partial(hook, old_impl=<older chain>).
    Returns the same chain with its innermost callable replaced by ``inner``."""
    from functools import partial
    if isinstance(chain, partial) and "old_impl" in chain.keywords:
        kw = dict(chain.keywords)
        kw["old_impl"] = _splice_innermost(kw["old_impl"], inner)
        return partial(chain.func, *chain.args, **kw)
    return inner


class HipBackend:
    """The product backend: HIP kernels through the C ABI.  ``engine`` is a scail_amd.dit.DiffusionTransformer whose
    parameters are (or share storage with) the host network's."""

    def __init__(self, engine=None):
        self.engine = engine
        self._cstep = None
        self._cond_src = None          # the (text, clip) tensors the cached K / V were projected from (strong references)
        self._cond = None

    # -- seam B3 -------------------------------------------------------------------------------
    def flash_attn_bhld(self, q, k, v, scale):
        """q (B, H, Lq, 128), k / v (B, H, Lk, 128) bf16 on the GPU -> context (B, H, Lq, 128) (a view of a (B, Lq, H, 128)
        buffer: the reference's following ``permute(0, 2, 1, 3).contiguous()`` (dit...:1094) is then free)."""
        from . import ops
        b, h, lq, d = q.shape
        tok = lambda t: t.permute(0, 2, 1, 3).reshape(b, t.shape[2], h * d)      # (B, L, H*128): the kernel's layout
        vt = ops.transpose_v(tok(v), h)
        o = ops.flash_attn(tok(q), tok(k), vt, scale=scale)
        return o.view(b, lq, h, d).permute(0, 2, 1, 3)

    # -- seam B2 -------------------------------------------------------------------------------
    def block(self, layer_id: int, hidden, mod, text, clip, rope, cond_key=None):
        """hidden (B, L, D) bf16 contiguous (updated in place and returned); mod (B, 6D) fp32; text (B, Lt, D), clip
        (Bc, Lc, D) bf16 = the EMBEDDED conditioning; rope = (T, Hp, Wp, H_shift, W_shift).  ``cond_key``: the (text, clip)
        OBJECTS the host network hands every layer of this forward (they differ from ``text`` / ``clip`` when the hook had to
        convert the dtype); default: the tensors themselves."""
        from .cstep import CStep
        eng = self.engine
        # The reference hands every layer of ONE forward the same encoder_outputs / image_clip_features and builds fresh ones per
        # forward (dit...:1505-1515).  Project K / V once per forward: always at layer 0 (every forward starts there), and whenever
        # the tensors are not the very objects the cache was built from.  The cache holds strong references, so the allocator
        # cannot hand the same address to a later forward's conditioning while the entry is alive (address + _version alone would
        # match a freed-and-reallocated tensor of another prompt).
        kt, kc = cond_key if cond_key is not None else (text, clip)
        src = self._cond_src
        if layer_id == 0 or src is None or src[0] is not kt or src[1] is not kc or src[2] != (kt._version, kc._version):
            self._cond = eng.kv_conditioning(text, clip)
            self._cond_src = (kt, kc, (kt._version, kc._version))
        if self._cstep is None:
            self._cstep = CStep(eng, eng.prepare())
        cos, sin = eng._rope(*rope, hidden.device)
        return self._cstep.block(layer_id, hidden, mod, self._cond, cos, sin)


def _build():
    BaseMixin, non_conflict = _sat_bases()

    class HipAttentionMixin(BaseMixin):
        """Seam B3: the ``attention_fn`` hook (signature sat/transformer_defaults.py:47-48)."""

        def __init__(self, backend=None):
            super().__init__()
            object.__setattr__(self, "backend", backend or HipBackend())

        @non_conflict
        def attention_fn(self, query_layer, key_layer, value_layer, attention_mask, attention_dropout=None,
                         log_attention_weights=None, scaling_attention_score=True, old_impl=None, **kwargs):
            # run the wrappers registered before this mixin (the rotary hook) around the HIP kernel instead of around SDPA
            chain = _splice_innermost(old_impl, self._hip_attention) if old_impl is not None else self._hip_attention
            return chain(query_layer, key_layer, value_layer, attention_mask, attention_dropout=attention_dropout,
                         log_attention_weights=log_attention_weights, scaling_attention_score=scaling_attention_score, **kwargs)

        def _hip_attention(self, query_layer, key_layer, value_layer, attention_mask, attention_dropout=None,
                           log_attention_weights=None, scaling_attention_score=True, **kwargs):
            """The innermost link: what attention_fn_default / UlyessAttentionMixin.attention_fn compute (unmasked SDPA)."""
            q, k, v = query_layer, key_layer, value_layer
            _require_single_sp_rank("attention_fn")
            if q.dim() != 4 or k.shape != v.shape or q.shape[:2] != k.shape[:2] or q.shape[3] != k.shape[3]:
                raise L.ScailHipError(f"attention_fn: expected q (B, heads, Lq, hd) and k, v (B, heads, Lk, hd), got "
                                      f"{tuple(q.shape)} {tuple(k.shape)} {tuple(v.shape)}")
            if q.shape[3] != 128:
                raise L.ScailHipError("attention_fn: the HIP flash attention is specialised for head_dim 128 (both shipped configs)")
            if attention_dropout is not None and getattr(attention_dropout, "training", False) and getattr(attention_dropout, "p", 0) > 0:
                raise L.ScailHipError("attention_fn: attention dropout in training mode is not on the sampling path")
            if log_attention_weights is not None:
                raise L.ScailHipError("attention_fn: log_attention_weights is not supported")
            if attention_mask is not None and attention_mask.numel() > 1 and not bool((attention_mask > 0).all()):
                raise L.ScailHipError("attention_fn: only the unmasked attention of the sampling path is implemented "
                                      "(sat/transformer_defaults.py:56 is_full branch)")
            scale = (q.shape[3] ** -0.5) if scaling_attention_score else 1.0
            return self.backend.flash_attn_bhld(q, k, v, scale)

    class HipLayerMixin(BaseMixin):
        """Seam B2: the ``layer_forward`` hook (AdaLNMixin.layer_forward, dit...:1009-1051; called from
        sat/model/transformer.py:712-719 with layer_id, the forward kwargs of dit...:1560-1586 and SAT's bookkeeping)."""

        def __init__(self, backend):
            super().__init__()
            object.__setattr__(self, "backend", backend)

        @non_conflict
        def layer_forward(self, hidden_states, mask, *args, old_impl=None, **kwargs):
            layer_id = int(kwargs["layer_id"])
            _require_single_sp_rank("layer_forward")
            emb = kwargs["emb"]                                   # adaln_projection(emb): (B, 6D)   dit...:1555, 1562
            text = kwargs["encoder_outputs"]                      # text_embedding(context): (B, Lt, D)   dit...:1505, 1563
            clip = kwargs["image_clip_features"]                  # clip_proj(...) repeated to B: (B, 257, D)  dit...:1507-1515
            B, Ltok, D = hidden_states.shape
            if emb.shape != (B, 6 * D):
                raise L.ScailHipError(f"layer_forward: emb must be (B, 6D) (share_adaln), got {tuple(emb.shape)}")
            table = self.backend_table(layer_id, emb.device)       # (1, 6, D) -> (6D) fp32
            mod = (emb.float() + table).contiguous()               # dit...:1025-1028
            rope = (int(kwargs["rope_T"]), int(kwargs["rope_H"]), int(kwargs["rope_W"]),
                    int(kwargs.get("rope_H_shift", 0)), int(kwargs.get("rope_W_shift", 0)))
            want = (1 + rope[0]) * rope[1] * rope[2] + rope[0] * (rope[1] // 2) * (rope[2] // 2)
            if Ltok != want:
                raise L.ScailHipError(f"layer_forward: {Ltok} tokens do not match [ref | noise | pose] of rope grid {rope[:3]} ({want})")
            dt = hidden_states.dtype
            h = hidden_states.to(torch.bfloat16).contiguous()
            if h.data_ptr() == hidden_states.data_ptr():
                h = h.clone()                                      # hooks return a new tensor; the kernel works in place
            as_bf16 = lambda t: t.to(torch.bfloat16).contiguous()
            out = self.backend.block(layer_id, h, mod, as_bf16(text), as_bf16(clip), rope, cond_key=(text, clip))
            return out.to(dt)

        def backend_table(self, layer_id, device):
            eng = self.backend.engine
            return eng.prepare()["adaln_tables"][layer_id] if eng is not None else None

    return HipAttentionMixin, HipLayerMixin


HipAttentionMixin, HipLayerMixin = _build()


def engine_from_reference(ref_net, share_weights: bool = True):
    """A scail_amd.dit.DiffusionTransformer over the parameters of a live reference ``DiffusionTransformer``
    (dit...:1209-1321): same hyper-parameters, ``load_state_dict(strict=True)`` of the reference's state dict;
    ``share_weights`` attaches the reference's own tensors (``assign=True``: no second copy of a 14B model)."""
    from .dit import DiffusionTransformer
    g = lambda n, d=None: getattr(ref_net, n, d)
    eng = DiffusionTransformer(
        transformer_args=dict(model_parallel_size=1, is_decoder=True), num_frames=g("num_frames"),
        time_compressed_rate=g("time_compressed_rate"), latent_width=g("latent_width"), latent_height=g("latent_height"),
        patch_size=tuple(g("patch_size")), in_channels=g("in_channels"), out_channels=g("out_channels"),
        hidden_size=g("hidden_size"), text_dim=g("text_dim", 4096), num_layers=g("num_layers"),
        num_attention_heads=g("num_attention_heads"),
        time_freq_dim=g("time_freq_dim"), time_embed_dim=g("time_embed_dim"), share_adaln=g("share_adaln"),
        inner_hidden_size=g("inner_hidden_size"), use_i2v_clip=g("use_i2v_clip"),
        layernorm_epsilon=g("layernorm_epsilon", 1e-6), init_seed=None)
    sd = ref_net.state_dict()
    missing, unexpected = eng.load_state_dict(sd, strict=True, assign=share_weights)
    assert not missing and not unexpected
    return eng


def install(ref_net, seam: str = "block", backend=None):
    """Hook the HIP path into a live reference network.  seam = "block" (B2: layer_forward) or "attention" (B3:
    attention_fn).  Returns the mixin.  ``backend`` defaults to HipBackend over the reference's own parameters."""
    if not hasattr(ref_net, "add_mixin"):
        raise TypeError("install() needs a SAT BaseModel (sat/model/base_model.py:80); for a config-level swap use "
                        "network_config.target: scail_amd.dit.DiffusionTransformer instead (seam B1)")
    # SAT may have been imported after this module: rebuild the classes on the real BaseMixin (add_mixin asserts isinstance)
    global HipAttentionMixin, HipLayerMixin
    real = sys.modules.get("sat.model.base_model")
    if real is not None and not issubclass(HipLayerMixin, real.BaseMixin):
        HipAttentionMixin, HipLayerMixin = _build()
    if seam == "attention":
        mix = HipAttentionMixin(backend)
        ref_net.add_mixin("hip_attention", mix)
    elif seam == "block":
        if backend is None:
            backend = HipBackend(engine_from_reference(ref_net))
        mix = HipLayerMixin(backend)
        ref_net.add_mixin("hip_layer", mix)
    else:
        raise ValueError(f"unknown seam {seam!r}: 'block' or 'attention'")
    return mix

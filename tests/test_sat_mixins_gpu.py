"""GPU side of the SAT seam (scail_amd/sat_mixins.py): the hook functions fed, through the real HipBackend (C ABI), the
same tensors SAT hands them inside the reference network -- the argument contract tests/test_sat_mixins_cpu.py pins against
the real reference's add_mixin / collect_hooks_ machinery in the build container (SAT itself is absent on the GPU box, so
the hooks are called directly with the kwargs of sat/model/transformer.py:712-719 + dit_video_crossattn_sc_xc.py:1560-1586)."""
import os

import numpy as np
import pytest
import torch

from oracle import scail_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _load(golden_dir, name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def test_attention_fn_hook_through_hip_backend():
    from scail_amd import sat_mixins
    mix = sat_mixins.HipAttentionMixin()
    g = torch.Generator().manual_seed(0)
    B, H, Lq, Lk = 2, 2, 300, 77
    bf = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16)
    q, k, v = bf(B, H, Lq, 128), bf(B, H, Lk, 128), bf(B, H, Lk, 128)
    mask = torch.ones(1, 1)
    out = mix.attention_fn(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV), None, cross_attention=True,
                           mem_cross=None, layer_id=torch.tensor(0))
    assert out.shape == (B, H, Lq, 128)
    want = O.sdpa(q.float(), k.float(), v.float())
    torch.testing.assert_close(out.float().cpu(), want, rtol=2e-2, atol=1e-2)
    # the reference's next op (dit...:1094) must be a free view
    assert out.permute(0, 2, 1, 3).is_contiguous()
    from scail_amd import lib as L
    with pytest.raises(L.ScailHipError):
        mix.attention_fn(q, k, v, mask)                                     # CPU tensors: no fallback
    with pytest.raises(L.ScailHipError, match="unmasked"):
        mix.attention_fn(q.to(DEV), k.to(DEV), v.to(DEV), torch.zeros(Lq, Lk, device=DEV))


def test_layer_forward_hook_through_hip_backend(golden_dir):
    """layer_forward with the kwargs SAT passes: equals the oracle block on the reference golden's inputs, layer by layer."""
    from scail_amd import sat_mixins
    from scail_amd.dit import DiffusionTransformer
    g = _load(golden_dir, "dit_tiny.npz")
    cfg = O.DiTConfig(**O.TINY)
    sd = O.make_state_dict(cfg, seed=int(g["seed"]))
    eng = DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=cfg.num_frames, latent_width=cfg.latent_width,
                               latent_height=cfg.latent_height, hidden_size=cfg.hidden_size, text_dim=cfg.text_dim,
                               num_layers=cfg.num_layers, num_attention_heads=cfg.num_attention_heads,
                               time_freq_dim=cfg.time_freq_dim, time_embed_dim=cfg.time_embed_dim, share_adaln=True,
                               inner_hidden_size=cfg.inner_hidden_size, use_i2v_clip=True, device=DEV)
    eng.load_state_dict(sd, strict=True)
    mix = sat_mixins.HipLayerMixin(sat_mixins.HipBackend(eng))
    # what the reference's forward computes before the layer loop (dit...:1505-1563), here by the oracle in fp32
    _, oh = O.dit_forward(cfg, sd, g["x"], g["t"], g["ctx"], g["ref"], g["pose"], g["clip"], return_hidden=True)
    text = O.text_embedding(cfg, sd, g["ctx"])
    clip = O.clip_proj(cfg, sd, g["clip"]).repeat(2, 1, 1)
    _, adaln = O.time_embeddings(cfg, sd, g["t"])
    B, T, _, H, W = g["x"].shape
    kw = dict(emb=adaln.to(DEV).to(torch.bfloat16), encoder_outputs=text.to(DEV).to(torch.bfloat16),
              image_clip_features=clip.to(DEV).to(torch.bfloat16), cross_attention_mask=torch.ones(B, text.shape[1], device=DEV),
              rope_T=T, rope_H=H // 2, rope_W=W // 2, rope_H_shift=0, rope_W_shift=0, seq_length=T * (H // 2) * (W // 2),
              ref_length=(H // 2) * (W // 2), pose_length=T * (H // 4) * (W // 4), position_ids=torch.ones(1, 1, device=DEV),
              output_this_layer={}, output_cross_layer={})
    for i in range(cfg.num_layers):
        hin = oh[i].to(DEV).to(torch.bfloat16)
        keep = hin.clone()
        out = mix.layer_forward(hin, torch.ones(1, 1, device=DEV), layer_id=torch.tensor(i), **kw)
        assert torch.equal(hin, keep), "the hook must not modify its input"
        assert out.dtype == torch.bfloat16 and out.shape == hin.shape
        torch.testing.assert_close(out.float().cpu(), oh[i + 1], rtol=2e-2, atol=2e-2, msg=lambda m: f"layer {i}: {m}")
        torch.testing.assert_close(out.float().cpu(), g[f"hidden{i + 1}"], rtol=2e-2, atol=2e-2)     # the reference's own
    from scail_amd import lib as L
    with pytest.raises(L.ScailHipError, match="tokens do not match"):
        mix.layer_forward(oh[0][:, :-1].to(DEV).to(torch.bfloat16), None, layer_id=torch.tensor(0), **kw)


def test_layer_forward_hook_second_conditioning_is_not_served_from_the_cache(golden_dir):
    """the reference builds fresh encoder_outputs / image_clip_features every forward (dit...:1505-1515); after they are
    freed the caching allocator commonly hands the same address (and _version 0) to the next forward's tensors.  A second prompt
    must be projected again, not rendered with the first prompt's cached K / V."""
    from scail_amd import sat_mixins
    from scail_amd.dit import DiffusionTransformer
    g = _load(golden_dir, "dit_tiny.npz")
    cfg = O.DiTConfig(**O.TINY)
    sd = O.make_state_dict(cfg, seed=int(g["seed"]))
    eng = DiffusionTransformer(transformer_args=dict(model_parallel_size=1), num_frames=cfg.num_frames, latent_width=cfg.latent_width,
                               latent_height=cfg.latent_height, hidden_size=cfg.hidden_size, text_dim=cfg.text_dim,
                               num_layers=cfg.num_layers, num_attention_heads=cfg.num_attention_heads,
                               time_freq_dim=cfg.time_freq_dim, time_embed_dim=cfg.time_embed_dim, share_adaln=True,
                               inner_hidden_size=cfg.inner_hidden_size, use_i2v_clip=True, device=DEV)
    eng.load_state_dict(sd, strict=True)
    mix = sat_mixins.HipLayerMixin(sat_mixins.HipBackend(eng))
    _, adaln = O.time_embeddings(cfg, sd, g["t"])
    B, T, _, H, W = g["x"].shape
    clip = O.clip_proj(cfg, sd, g["clip"]).repeat(2, 1, 1)
    cos, sin = O.rope_tables(cfg, T, H // 2, W // 2)
    _, oh = O.dit_forward(cfg, sd, g["x"], g["t"], g["ctx"], g["ref"], g["pose"], g["clip"], return_hidden=True)
    hin = oh[0]
    ptrs = []
    for trial, ctx in enumerate((g["ctx"], torch.flip(g["ctx"], dims=[1]) * 1.5)):
        text = O.text_embedding(cfg, sd, ctx)
        want = O.block(cfg, sd, 0, hin, adaln, text, clip, cos, sin)
        text_g, clip_g = text.to(DEV).to(torch.bfloat16), clip.to(DEV).to(torch.bfloat16)      # fresh tensors, as in a new forward
        ptrs.append(text_g.data_ptr())
        kw = dict(emb=adaln.to(DEV).to(torch.bfloat16), encoder_outputs=text_g, image_clip_features=clip_g,
                  rope_T=T, rope_H=H // 2, rope_W=W // 2, rope_H_shift=0, rope_W_shift=0)
        out = mix.layer_forward(hin.to(DEV).to(torch.bfloat16), None, layer_id=torch.tensor(0), **kw)
        torch.testing.assert_close(out.float().cpu(), want, rtol=2e-2, atol=2e-2, msg=lambda m: f"conditioning {trial}: {m}")
        del text_g, clip_g, kw, out
    print("conditioning tensor addresses of the two forwards:", ptrs)

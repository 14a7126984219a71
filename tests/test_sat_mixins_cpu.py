"""The SAT seam, executed inside the REFERENCE's own network (build container only: needs /root/reference).

scail_amd.sat_mixins hooks ``layer_forward`` (seam B2) / ``attention_fn`` (seam B3) into the reference
``DiffusionTransformer`` through ``add_mixin`` (sat/model/base_model.py:114-176).  Here the real reference network is
built on CPU (oracle/ref_shims.py), the mixins are installed with a stand-in backend that (1) asserts the documented
argument contract of the HIP binding and (2) evaluates the oracle's fp32 formula for the same call, and the network's
output is compared with the reference's own golden output -- so the hook table resolves, the hooks are actually the ones
SAT calls, and the tensors arrive with the shapes / meaning the GPU binding expects.  The GPU side of the same contract
(the same calls through HipBackend) is tests/test_sat_mixins_gpu.py."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_shims
from oracle import scail_oracle as O

pytestmark = pytest.mark.skipif(not ref_shims.available(), reason="needs the reference tree (build container only)")


def _load(golden_dir, name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _forward(net, g):
    with torch.no_grad():
        return net(g["x"], timesteps=g["t"], context=g["ctx"], concat_images=torch.zeros(1, *g["x"].shape[1:]),
                   ref_concat=g["ref"], concat_smpl_render=g["pose"], image_clip_features=g["clip"])


class _AttentionStandIn:
    """Contract of HipBackend.flash_attn_bhld + the oracle formula (sat/transformer_defaults.py:67-72)."""

    def __init__(self, heads):
        self.heads, self.calls = heads, []

    def flash_attn_bhld(self, q, k, v, scale):
        assert q.dim() == 4 and q.shape[1] == self.heads and q.shape[3] == 128
        assert k.shape == v.shape and k.shape[:2] == q.shape[:2] and k.shape[3] == 128
        assert abs(scale - 128 ** -0.5) < 1e-9
        self.calls.append((q.shape[2], k.shape[2]))
        # the HIP binding's layout round trip: (B, H, L, d) -> (B, L, H*d) -> kernel -> (B, Lq, H, d) viewed as (B, H, Lq, d)
        b, h, lq, d = q.shape
        tok = lambda t: t.permute(0, 2, 1, 3).reshape(b, t.shape[2], h * d)
        o = O._merge(O.sdpa(O._heads(tok(q), h), O._heads(tok(k), h), O._heads(tok(v), h)))
        return o.view(b, lq, h, d).permute(0, 2, 1, 3)


def test_attention_fn_mixin_inside_reference_network(golden_dir):
    from scail_amd import sat_mixins
    g = _load(golden_dir, "dit_tiny.npz")
    cfg = O.DiTConfig(**O.TINY)
    net = ref_shims.build_reference_dit(cfg, O.make_state_dict(cfg, seed=int(g["seed"])))
    be = _AttentionStandIn(cfg.num_attention_heads)
    mix = sat_mixins.install(net, seam="attention", backend=be)
    assert isinstance(mix, ref_shims.load_reference()["dit"].BaseMixin)
    assert net.hook_origins["attention_fn"].startswith("hip_attention"), net.hook_origins       # stacked on 'ulysse'
    out = _forward(net, g)
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-4)
    # per layer: self-attention (L x L), text cross-attention (L x Lt), CLIP cross-attention (L x Lc)
    L = g["hidden1"].shape[1]
    assert be.calls == [(L, L), (L, g["ctx"].shape[1]), (L, g["clip"].shape[1])] * cfg.num_layers
    net.del_mixin("hip_attention")
    assert not net.hook_origins["attention_fn"].startswith("hip_attention")


class _BlockStandIn:
    """Contract of HipBackend.block + the oracle block (dit...:1009-1051) on the same arguments."""

    def __init__(self, cfg, sd):
        self.cfg, self.sd, self.calls, self.engine = cfg, sd, [], None

    def block(self, layer_id, hidden, mod, text, clip, rope, cond_key=None):
        cfg = self.cfg
        assert cond_key is not None and len(cond_key) == 2          # the host network's own (text, clip) objects
        B, Ltok, D = hidden.shape
        assert hidden.dtype == torch.bfloat16 and hidden.is_contiguous()
        assert mod.shape == (B, 6 * D) and mod.dtype == torch.float32 and mod.is_contiguous()
        assert text.dtype == torch.bfloat16 and text.shape[0] == B and text.shape[2] == D
        assert clip.dtype == torch.bfloat16 and clip.shape[0] == B and clip.shape[2] == D
        T, Hp, Wp, hs, ws = rope
        assert Ltok == (1 + T) * Hp * Wp + T * (Hp // 2) * (Wp // 2) and (hs, ws) == (0, 0)
        self.calls.append(layer_id)
        cos, sin = O.rope_tables(cfg, T, Hp, Wp, hs, ws)
        # mod already contains the layer's table: hand O.block the embedding part only
        emb = mod - self.sd[f"mixins.adaln_layer.adaLN_modulations.{layer_id}"].reshape(1, 6 * D)
        out = O.block(cfg, self.sd, layer_id, hidden.float(), emb, text.float(), clip.float(), cos, sin)
        return out.to(torch.bfloat16)


def test_layer_forward_mixin_inside_reference_network(golden_dir):
    from scail_amd import sat_mixins
    g = _load(golden_dir, "dit_tiny.npz")
    cfg = O.DiTConfig(**O.TINY)
    sd = O.make_state_dict(cfg, seed=int(g["seed"]))
    net = ref_shims.build_reference_dit(cfg, sd)
    be = _BlockStandIn(cfg, sd)
    mix = sat_mixins.install(net, seam="block", backend=be)
    # the stand-in has no engine: give the mixin the layer tables the HIP engine would hold
    mix.backend_table = lambda i, dev: sd[f"mixins.adaln_layer.adaLN_modulations.{i}"].reshape(6 * cfg.hidden_size).float()
    assert net.hook_origins["layer_forward"] == "hip_layer -> adaln_layer", net.hook_origins
    out = _forward(net, g)
    assert be.calls == list(range(cfg.num_layers))
    # hidden states cross the seam in bf16 (the kernel's storage type): bf16 tolerance against the fp32 golden
    torch.testing.assert_close(out, g["out"], rtol=2e-2, atol=2e-2)


def test_engine_from_reference_shares_the_reference_parameters(golden_dir):
    """engine_from_reference(): strict state-dict match, parameters attached by reference (no copy)."""
    from scail_amd import sat_mixins
    cfg = O.DiTConfig(**O.TINY)
    net = ref_shims.build_reference_dit(cfg, O.make_state_dict(cfg, seed=3))
    eng = sat_mixins.engine_from_reference(net)
    a = dict(net.named_parameters())["transformer.layers.1.mlp.dense_h_to_4h.weight"]
    b = dict(eng.named_parameters())["transformer.layers.1.mlp.dense_h_to_4h.weight"]
    assert a.data_ptr() == b.data_ptr() and tuple(a.shape) == tuple(b.shape)
    assert set(dict(eng.named_parameters())) == set(net.state_dict())
    with pytest.raises(TypeError):
        sat_mixins.install(torch.nn.Linear(2, 2))


def test_hooks_refuse_a_sequence_parallel_world(golden_dir, monkeypatch):
    """both seams replace code that exchanges between sequence-parallel ranks (UlyessAttentionMixin.attention_fn,
    dit...:351-379); with SAT's sequence-parallel world > 1 they must raise instead of attending to the local shard only."""
    import sys
    import types
    from scail_amd import sat_mixins
    from scail_amd import lib as L
    stub = types.ModuleType("sat.mpu")
    stub.get_sequence_parallel_world_size = lambda: 2
    monkeypatch.setitem(sys.modules, "sat.mpu", stub)
    assert sat_mixins._sequence_parallel_world_size() == 2
    q = torch.zeros(1, 2, 8, 128, dtype=torch.bfloat16)
    with pytest.raises(L.ScailHipError, match="sequence-parallel world size is 2"):
        sat_mixins.HipAttentionMixin(backend=object())._hip_attention(q, q, q, None)
    with pytest.raises(L.ScailHipError, match="sequence-parallel world size is 2"):
        sat_mixins.HipLayerMixin(backend=object()).layer_forward(torch.zeros(1, 4, 8), None, layer_id=0)
    stub.get_sequence_parallel_world_size = lambda: 1
    assert sat_mixins._sequence_parallel_world_size() == 1

    def boom():
        raise AssertionError("sequence parallel group is not initialized")      # sat/mpu/initialize.py asserts when unset
    stub.get_sequence_parallel_world_size = boom
    assert sat_mixins._sequence_parallel_world_size() == 1


def test_block_backend_conditioning_cache_is_per_forward():
    """HipBackend.block caches the per-layer cross-attention K / V of ONE forward.  A new forward (layer 0) or new
    conditioning objects must re-project, also when a freed tensor's address and _version come back (the old key)."""
    from scail_amd import sat_mixins

    class Eng:
        def __init__(self):
            self.n = 0

        def kv_conditioning(self, text, clip):
            self.n += 1
            return {"tag": float(text.flatten()[0])}

        def prepare(self):
            return {}

        def _rope(self, *a):
            return None, None

    class Step:
        def block(self, layer, hidden, mod, cond, cos, sin):
            return cond["tag"]

    be = sat_mixins.HipBackend(Eng())
    be._cstep = Step()
    h = torch.zeros(1, 2, 4)
    t1, c1 = torch.full((1, 2, 4), 1.0), torch.zeros(1, 2, 4)
    assert [be.block(i, h, None, t1, c1, (1, 1, 1, 0, 0)) for i in range(3)] == [1.0, 1.0, 1.0] and be.engine.n == 1
    assert be.block(0, h, None, t1, c1, (1, 1, 1, 0, 0)) == 1.0 and be.engine.n == 2          # next forward: layer 0 re-projects
    t2 = torch.full((1, 2, 4), 2.0)                                                            # another prompt mid-stack
    assert be.block(1, h, None, t2, c1, (1, 1, 1, 0, 0)) == 2.0 and be.engine.n == 3
    # dtype-converted copies per layer, keyed on the host's own objects: no re-projection inside one forward
    assert be.block(2, h, None, t2.clone(), c1.clone(), (1, 1, 1, 0, 0), cond_key=(t2, c1)) == 2.0 and be.engine.n == 3
    t2.add_(1.0)                                                                               # in-place edit bumps _version
    assert be.block(3, h, None, t2, c1, (1, 1, 1, 0, 0)) == 3.0 and be.engine.n == 4

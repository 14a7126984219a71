"""Checkpoint ingest: the tensor-parallel merge of scail_amd/checkpoint.py.
  * partition -> merge is the identity on a DiT state dict and save_checkpoint(model_parallel_size=n) / load_checkpoint
    round-trips through the reference's directory layout (sat/training/model_io.py:36-48);
  * (build container) the partition rule equals the REAL reference's ColumnParallelLinear.partition /
    RowParallelLinear.partition (sat/mpu/layers.py:286-340, 438-460) for the strides the DiT uses."""
import pytest
import torch

from oracle import ref_shims
from oracle import scail_oracle as O


def test_partition_merge_roundtrip_and_files(tmp_path):
    from scail_amd import checkpoint as C
    cfg = O.DiTConfig(**O.TINY)
    sd = {"model.diffusion_model." + k: v for k, v in O.make_state_dict(cfg, seed=5).items()}
    for n in (2, 4):
        parts = C.partition_state_dict(sd, n)
        k = "model.diffusion_model.transformer.layers.0.attention.query_key_value.weight"
        assert parts[0][k].shape == (3 * cfg.hidden_size // n, cfg.hidden_size)
        k2 = "model.diffusion_model.transformer.layers.1.mlp.dense_4h_to_h.weight"
        assert parts[0][k2].shape == (cfg.hidden_size, cfg.inner_hidden_size // n)
        # rank 1 of the fused qkv holds q[1], k[1], v[1] slices -- not a contiguous third
        D = cfg.hidden_size
        sub = D // n
        assert torch.equal(parts[1][k], torch.cat([sd[k][c * D + sub:c * D + 2 * sub] for c in range(3)]))
        merged = C.merge_model_parallel_state_dicts(parts)
        assert set(merged) == set(sd) and all(torch.equal(merged[q], sd[q]) for q in sd)
    C.save_checkpoint(sd, str(tmp_path), 7, model_parallel_size=2)
    assert len(C.model_parallel_files(str(tmp_path), 7, False)) == 2

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            from scail_amd.dit import _register
            for name, t in sd.items():
                _register(self, name, torch.nn.Parameter(torch.zeros_like(t), requires_grad=False))

    h = Holder()
    it, missing, unexpected = C.load_checkpoint(h, str(tmp_path))
    assert it == 7 and not missing and not unexpected
    got = h.state_dict()
    assert all(torch.equal(got[q], sd[q]) for q in sd)
    parts = C.partition_state_dict(sd, 2)
    parts[1]["model.diffusion_model.time_embed.0.weight"] = parts[1]["model.diffusion_model.time_embed.0.weight"] + 1
    with pytest.raises(ValueError, match="replicated parameter"):
        C.merge_model_parallel_state_dicts(parts)


@pytest.mark.skipif(not ref_shims.available(), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("stride,name", [(3, "transformer.layers.0.attention.query_key_value"),
                                         (2, "transformer.layers.0.cross_attention.key_value"),
                                         (2, "mixins.adaln_layer.clip_feature_key_value_list.0"),
                                         (1, "transformer.layers.0.mlp.dense_h_to_4h")])
def test_column_partition_rule_equals_reference(stride, name):
    from scail_amd import checkpoint as C
    ref_shims.load_reference()
    from sat import mpu
    if not mpu.model_parallel_is_initialized():
        mpu.initialize_model_parallel(1)
    from sat.mpu.layers import ColumnParallelLinear
    g = torch.Generator().manual_seed(1)
    out_f, in_f = 24 * stride, 16
    lin = ColumnParallelLinear(in_f, out_f, stride=stride, gather_output=False, bias=True)
    w = torch.randn(out_f, in_f, generator=g)
    for n in (2, 4):
        want = lin.partition(new_model_parallel_size=n, full_weight=w)
        got = C.partition_state_dict({name + ".weight": w}, n)
        for r in range(n):
            assert torch.equal(got[r][name + ".weight"], want[r])


@pytest.mark.skipif(not ref_shims.available(), reason="needs the reference tree (build container only)")
def test_row_partition_rule_equals_reference():
    from scail_amd import checkpoint as C
    ref_shims.load_reference()
    from sat import mpu
    if not mpu.model_parallel_is_initialized():
        mpu.initialize_model_parallel(1)
    from sat.mpu.layers import RowParallelLinear
    lin = RowParallelLinear(32, 8, input_is_parallel=True, bias=True)
    w = torch.randn(8, 32, generator=torch.Generator().manual_seed(2))
    want = lin.partition(new_model_parallel_size=4, full_weight=w)
    got = C.partition_state_dict({"transformer.layers.3.attention.dense.weight": w,
                                  "transformer.layers.3.attention.dense.bias": torch.ones(8)}, 4)
    for r in range(4):
        assert torch.equal(got[r]["transformer.layers.3.attention.dense.weight"], want[r])
        assert torch.equal(got[r]["transformer.layers.3.attention.dense.bias"], torch.ones(8))

"""scail_amd/conditioner.py: the reference's GeneralConditioner protocol (modules.py:86-244) and get_batch
(sample_video.py:109-178) with stand-in embedders (the real ones run on the GPU: tests/test_encoders_gpu.py)."""
import sys
import types

import pytest
import torch
from torch import nn

from scail_amd import conditioner as C
from scail_amd.config import get_obj_from_str


class _Txt(nn.Module):
    def __init__(self, width=4):
        super().__init__()
        self.width = width

    def forward(self, texts):
        return torch.stack([torch.full((3, self.width), float(len(t))) for t in texts])          # (B, 3, width) -> crossattn


class _Vec(nn.Module):
    def forward(self, a, b):
        return [a.float()[:, None] + b.float()[:, None], torch.ones(a.shape[0], 2, 5, 5)]        # vector + concat


@pytest.fixture()
def stub_module():
    m = types.ModuleType("_stub_embedders")
    m.Txt, m.Vec = _Txt, _Vec
    sys.modules["_stub_embedders"] = m
    yield
    del sys.modules["_stub_embedders"]


def test_protocol(stub_module):
    assert get_obj_from_str("sgm.modules.GeneralConditioner") is C.GeneralConditioner
    cond = C.GeneralConditioner([
        {"target": "_stub_embedders.Txt", "params": {"width": 4}, "input_key": "txt", "ucg_rate": 0.1, "legacy_ucg_val": ""},
        {"target": "_stub_embedders.Txt", "params": {"width": 2}, "input_key": "txt2"},
        {"target": "_stub_embedders.Vec", "input_keys": ["a", "b"]}])
    assert sorted(C.get_unique_embedder_keys_from_conditioner(C.GeneralConditioner([
        {"target": "_stub_embedders.Txt", "input_key": "txt"}]))) == ["txt"]
    batch, batch_uc = C.get_batch(["txt"], {"prompt": "a girl", "negative_prompt": "", "num_frames": torch.tensor([21])}, [2], device="cpu")
    assert batch == {"txt": ["a girl", "a girl"]} and batch_uc == {"txt": ["", ""]}
    extra = {"txt2": ["xy", "xy"], "a": torch.tensor([1, 2]), "b": torch.tensor([10, 20])}
    batch.update(extra)
    batch_uc.update(extra)
    c, uc = cond.get_unconditional_conditioning(batch, batch_uc, force_uc_zero_embeddings=["txt2"])
    assert set(c) == {"crossattn", "vector", "concat"}
    assert c["crossattn"].shape == (2, 3, 6)                                    # crossattn embeddings concatenate on dim 2
    assert torch.equal(c["crossattn"][..., :4], torch.full((2, 3, 4), 6.0)) and torch.equal(c["crossattn"][..., 4:], torch.full((2, 3, 2), 2.0))
    assert torch.equal(uc["crossattn"][..., :4], torch.zeros(2, 3, 4))          # "" has length 0
    assert torch.equal(uc["crossattn"][..., 4:], torch.zeros(2, 3, 2))          # forced to zero for the unconditional branch only
    assert torch.equal(c["vector"], torch.tensor([[11.0], [22.0]])) and c["concat"].shape == (2, 2, 5, 5)
    assert not any(p.requires_grad for p in cond.parameters())
    with pytest.raises(KeyError, match="input_key"):
        C.GeneralConditioner([{"target": "_stub_embedders.Txt"}])
    with pytest.raises(NotImplementedError, match="training"):
        C.GeneralConditioner([{"target": "_stub_embedders.Txt", "input_key": "txt", "is_trainable": True}])

"""GPU parity tests for the conditioning encoders (UMT5 text encoder, CLIP visual tower) against outputs of the
REAL reference classes (tests/golden/encoders_tiny.npz) and the CPU oracle.  bf16 tolerance 3e-2."""
import os

import numpy as np
import pytest
import torch

from oracle import encoders_oracle as E

pytestmark = pytest.mark.gpu
DEV = "cuda"


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "encoders_tiny.npz"))
    out = {}
    for k in z.files:
        a = torch.from_numpy(np.asarray(z[k]))
        out[k] = a.view(torch.bfloat16).float() if a.dtype == torch.int16 else a
    return out


def split(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def test_attn_small_bias_mask_and_head80():
    from scail_amd import ops
    from scail_amd.umt5 import relative_position_bucket
    g = torch.Generator().manual_seed(0)
    B, H, Lq, hd = 2, 3, 50, 80
    q, k, v = (torch.randn(B, Lq, H * hd, generator=g).to(torch.bfloat16) for _ in range(3))
    tab = torch.randn(32, H, generator=g)
    bucket = relative_position_bucket(Lq, Lq)
    assert torch.equal(bucket.long(), E.t5_bucket(Lq, Lq))
    mask = torch.ones(B, Lq, dtype=torch.int32)
    mask[1, 30:] = 0
    hv = lambda t: t.float().view(B, Lq, H, hd).permute(0, 2, 1, 3)
    s = hv(q) @ hv(k).transpose(-1, -2) * 0.5 + tab[bucket.long()].permute(2, 0, 1)[None]
    s = s.masked_fill(mask.view(B, 1, 1, -1) == 0, float("-inf"))
    ref = (torch.softmax(s, -1) @ hv(v)).permute(0, 2, 1, 3).reshape(B, Lq, H * hd)
    o = ops.attn_small(q.to(DEV), k.to(DEV), v.to(DEV), H, scale=0.5, bucket=bucket.to(DEV), bias_tab=tab.to(DEV), key_mask=mask.to(DEV))
    torch.testing.assert_close(o.float().cpu(), ref, rtol=2e-2, atol=2e-2)


def test_t5_encoder_vs_reference_golden(golden_dir):
    from scail_amd.umt5 import T5Encoder
    g = load(golden_dir)
    sd = split(g, "t5.")
    enc = T5Encoder(vocab=100, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=2, num_buckets=32, shared_pos=False, device=DEV)
    missing, unexpected = enc.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    out = enc(g["ids"].to(DEV), g["mask"].to(DEV)).float().cpu()
    valid = g["mask"].bool()
    torch.testing.assert_close(out[valid], g["t5_out"][valid], rtol=3e-2, atol=3e-2)       # padded rows are zeroed downstream
    torch.testing.assert_close(out, g["t5_out"], rtol=3e-2, atol=3e-2)


def test_clip_visual_vs_reference_golden(golden_dir):
    from scail_amd.clip import VisionTransformer
    g = load(golden_dir)
    sd = split(g, "vit.")
    vit = VisionTransformer(image_size=56, patch_size=14, dim=192, mlp_ratio=4, out_dim=64, num_heads=2, num_layers=3, device=DEV)
    missing, unexpected = vit.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    out = vit(g["imgs"].to(DEV), use_31_block=True).float().cpu()
    torch.testing.assert_close(out, g["vit_out"], rtol=3e-2, atol=3e-2)


def test_encoders_full_size_shapes_and_wrappers():
    """Full-size towers with random weights: shapes the DiT consumes -- text (B,512,4096) with zeroed padding rows,
    CLIP (B,257,1280) -- through the conditioner-style wrappers."""
    from scail_amd.clip import CLIPModel
    from scail_amd.umt5 import T5EncoderModel
    t5 = T5EncoderModel(max_length=512, device=DEV, num_layers=2)                  # 2 of 24 layers: same kernels and shapes
    ids = torch.randint(0, 256384, (1, 512))
    mask = torch.zeros(1, 512, dtype=torch.long)
    mask[:, :37] = 1
    z = t5(ids.to(DEV), mask.to(DEV))
    assert z.shape == (1, 512, 4096) and torch.isfinite(z.float()).all() and float(z[0, 37:].abs().max()) == 0 and float(z[0, :37].abs().max()) > 0
    clip = CLIPModel(device=DEV, num_layers=3)
    feats = clip.visual([torch.rand(3, 1, 512, 896, device=DEV) * 2 - 1])
    assert feats.shape == (1, 257, 1280) and torch.isfinite(feats.float()).all()


def test_text_conditioner_from_strings(tmp_path):
    """T5EncoderModel.encode_text (reference ``__call__(texts)``, umt5.py:512-535): tokeniser -> encoder -> padded rows zeroed,
    varlen cut.  Tiny encoder, SentencePiece model trained here (no tokenizer files ship offline)."""
    import io
    spm = pytest.importorskip("sentencepiece")
    from scail_amd.umt5 import T5EncoderModel
    corpus = ["the girl is dancing in the street", "a man walks his dog", "two people are dancing together"] * 30
    buf = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(corpus), model_writer=buf, vocab_size=40, model_type="unigram",
                                   hard_vocab_limit=False, minloglevel=2, pad_id=0, eos_id=1, unk_id=2, bos_id=-1)
    (tmp_path / "spiece.model").write_bytes(buf.getvalue())
    kw = dict(vocab=64, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=2, num_buckets=32, shared_pos=False)
    m = T5EncoderModel(max_length=16, device=DEV, tokenizer_path=str(tmp_path / "spiece.model"), **kw)
    z = m.encode_text(["the girl is dancing", ""])
    ids, mask = m.tokenizer(["the girl is dancing", ""], return_mask=True)
    assert z.shape == (2, 16, 128) and torch.isfinite(z.float()).all()
    assert (z.float().cpu()[mask == 0] == 0).all() and (z.float().cpu()[mask == 1].abs().sum(-1) > 0).all()
    torch.testing.assert_close(z, m(ids.to(DEV), mask.to(DEV)))
    v = T5EncoderModel(max_length=16, device=DEV, tokenizer_path=str(tmp_path / "spiece.model"), varlen_text=True,
                       cond_length_multiple=4, uncond_text_length=3, **kw)
    n = int(mask[0].sum())
    assert v.encode_text("the girl is dancing").shape[1] == (n + 3) // 4 * 4 and v.encode_text("").shape[1] == 3
    with pytest.raises(RuntimeError, match="tokenizer_path"):
        T5EncoderModel(max_length=16, device=DEV, **kw).encode_text("x")

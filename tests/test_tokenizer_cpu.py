"""scail_amd/tokenizer.py (reference sgm/modules/encoders/tokenizers.py): cleaning rules and the SentencePiece route with
T5 conventions, on a tiny unigram model trained inside the test (no tokenizer files ship offline)."""
import io

import pytest
import torch

from scail_amd import tokenizer as T


def test_cleaning_rules():
    assert T.whitespace_clean(T.basic_clean("  a \n\t b &amp;amp; c  ")) == "a b & c"
    assert T.canonicalize("Hello,_World!  It's") == "hello world its"
    assert T.canonicalize("a.b||c.d", keep_punctuation_exact_string="||") == "ab||cd"
    with pytest.raises(ValueError):
        T.HuggingfaceTokenizer("x", clean="bogus")
    with pytest.raises(FileNotFoundError, match="nothing is downloaded"):
        T.HuggingfaceTokenizer("/no/such/tokenizer", seq_len=8, clean="whitespace")


@pytest.fixture(scope="module")
def sp_model(tmp_path_factory):
    spm = pytest.importorskip("sentencepiece")
    corpus = ["the girl is dancing in the street", "a man walks his dog", "two people are dancing together",
              "the dog runs in the park", "a woman is singing a song"] * 20
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(corpus), model_writer=model, vocab_size=40, model_type="unigram", hard_vocab_limit=False, minloglevel=2,
                                   pad_id=0, eos_id=1, unk_id=2, bos_id=-1)
    p = tmp_path_factory.mktemp("sp") / "spiece.model"
    p.write_bytes(model.getvalue())
    return str(p)


def test_sentencepiece_route_follows_t5_conventions(sp_model):
    tok = T.HuggingfaceTokenizer(sp_model, seq_len=12, clean="whitespace")
    ids, mask = tok(["the girl   is dancing", ""], return_mask=True)
    assert ids.shape == (2, 12) and ids.dtype == torch.int64 and mask.dtype == torch.int64
    n = int(mask[0].sum())
    assert 2 <= n <= 12 and ids[0, n - 1] == 1 and (ids[0, n:] == 0).all() and (mask[0, n:] == 0).all()
    assert int(mask[1].sum()) == 1 and ids[1, 0] == 1                      # the empty prompt is just </s>
    assert torch.equal(tok("the girl is dancing"), ids[:1])                # whitespace cleaning, str input
    long = tok("the dog runs in the park " * 10, return_mask=True)
    assert long[0].shape == (1, 12) and long[0][0, -1] == 1 and int(long[1].sum()) == 12   # truncation keeps </s>
    free = T.HuggingfaceTokenizer(sp_model)(["a man", "a man walks his dog"])
    assert free.shape[0] == 2 and (free[0] == 0).any() and not (free[1] == 0).any()        # padded to the longest

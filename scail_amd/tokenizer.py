"""Text tokenisation in front of the UMT5 encoder (reference sgm/modules/encoders/tokenizers.py:38-82,
``HuggingfaceTokenizer(name, seq_len, clean='whitespace')`` as built by T5EncoderModel, umt5.py:509-510).

Two sources, chosen by what ``name`` points at:
  * a Hugging Face tokenizer directory (the reference's ``google/umt5-xxl`` files) -> ``transformers.AutoTokenizer``;
  * a raw SentencePiece ``*.model`` file -> ``sentencepiece`` directly with the T5 conventions the HF class applies
    (``</s>`` appended, right padding with ``<pad>``, truncation keeps room for ``</s>``).
Neither file set ships in this offline image, so construction fails loudly when ``name`` does not exist; nothing is
downloaded.  ``ftfy`` (mojibake repair inside ``basic_clean``) is not installed here: when absent, that step is skipped
and only the HTML un-escaping + whitespace normalisation of the reference remain.
"""
from __future__ import annotations

import html
import os
import re
import string
from typing import List, Sequence, Union

import torch

try:  # optional, like every other absent package of this image
    import ftfy  # type: ignore
except Exception:  # pragma: no cover
    ftfy = None


def basic_clean(text: str) -> str:
    if ftfy is not None:
        text = ftfy.fix_text(text)
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text: str) -> str:
    return re.sub(r"\s+", " ", text).strip()


def canonicalize(text: str, keep_punctuation_exact_string: str = None) -> str:
    strip = str.maketrans("", "", string.punctuation)
    text = text.replace("_", " ")
    if keep_punctuation_exact_string:
        text = keep_punctuation_exact_string.join(p.translate(strip) for p in text.split(keep_punctuation_exact_string))
    else:
        text = text.translate(strip)
    return re.sub(r"\s+", " ", text.lower()).strip()


class HuggingfaceTokenizer:
    def __init__(self, name: str, seq_len: int = None, clean: str = None, **kwargs):
        if clean not in (None, "whitespace", "lower", "canonicalize"):
            raise ValueError(f"clean must be None, 'whitespace', 'lower' or 'canonicalize', got {clean!r}")
        if name is None or not os.path.exists(name):
            raise FileNotFoundError(f"tokenizer files not found at {name!r} (a Hugging Face tokenizer directory or a SentencePiece "
                                    ".model file); nothing is downloaded")
        self.name, self.seq_len, self.clean = name, seq_len, clean
        self._sp = None
        if os.path.isfile(name):
            import sentencepiece as spm
            self._sp = spm.SentencePieceProcessor(model_file=name)
            self.vocab_size = self._sp.get_piece_size()
            self.eos_id = self._sp.eos_id() if self._sp.eos_id() >= 0 else 1
            self.pad_id = self._sp.pad_id() if self._sp.pad_id() >= 0 else 0
        else:
            from transformers import AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(name, local_files_only=True, **kwargs)
            self.vocab_size = self.tokenizer.vocab_size

    def _clean(self, text: str) -> str:
        if self.clean == "whitespace":
            return whitespace_clean(basic_clean(text))
        if self.clean == "lower":
            return whitespace_clean(basic_clean(text)).lower()
        if self.clean == "canonicalize":
            return canonicalize(basic_clean(text))
        return text

    def __call__(self, sequence: Union[str, Sequence[str]], return_mask: bool = False, add_special_tokens: bool = True, **kwargs):
        """-> ids (B, seq_len) int64 [, attention mask (B, seq_len) int64]."""
        if isinstance(sequence, str):
            sequence = [sequence]
        if self.clean:
            sequence = [self._clean(u) for u in sequence]
        if self._sp is None:
            kw = {"return_tensors": "pt", "add_special_tokens": add_special_tokens}
            if self.seq_len is not None:
                kw.update(padding="max_length", truncation=True, max_length=self.seq_len)
            kw.update(kwargs)
            out = self.tokenizer(list(sequence), **kw)
            return (out.input_ids, out.attention_mask) if return_mask else out.input_ids
        rows: List[List[int]] = []
        for u in sequence:
            ids = list(self._sp.encode(u))
            if self.seq_len is not None:
                ids = ids[:self.seq_len - (1 if add_special_tokens else 0)]
            if add_special_tokens:
                ids.append(self.eos_id)
            rows.append(ids)
        width = self.seq_len if self.seq_len is not None else max(len(r) for r in rows)
        ids_t = torch.full((len(rows), width), self.pad_id, dtype=torch.int64)
        mask = torch.zeros(len(rows), width, dtype=torch.int64)
        for i, r in enumerate(rows):
            ids_t[i, :len(r)] = torch.tensor(r, dtype=torch.int64)
            mask[i, :len(r)] = 1
        return (ids_t, mask) if return_mask else ids_t

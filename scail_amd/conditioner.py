"""Inference side of the reference's conditioner (sgm/modules/encoders/modules.py:86-244 ``GeneralConditioner``,
sample_video.py:105-178 ``get_batch`` / ``get_unique_embedder_keys_from_conditioner``): the object the CLI asks for
``c, uc = conditioner.get_unconditional_conditioning(batch, batch_uc, force_uc_zero_embeddings)`` (:433-438).

Built from the reference's own ``conditioner_config`` (config.TARGET_MAP resolves ``sgm.modules.GeneralConditioner`` and the
embedder targets).  Only what sampling needs is here: embedders are frozen, and the training-time condition dropout
(``ucg_rate`` Bernoulli masks, ``legacy_ucg_val`` random replacement, ``cor_embs``) is not -- ``forward`` is the
deterministic embedding, which is also what the reference computes inside ``get_unconditional_conditioning`` (it zeroes
every ``ucg_rate`` around the two calls, :228-243)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from .config import instantiate_from_config


class GeneralConditioner(nn.Module):
    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}

    def __init__(self, emb_models: Sequence[Dict], cor_embs=(), cor_p=(), embedder_kwargs: Optional[Dict] = None):
        super().__init__()
        embedders = []
        for cfg in emb_models:
            emb = instantiate_from_config(cfg, **(embedder_kwargs or {}))
            emb.is_trainable = cfg.get("is_trainable", False)
            emb.ucg_rate = cfg.get("ucg_rate", 0.0)
            emb.legacy_ucg_val = cfg.get("legacy_ucg_val", None)
            if "input_key" in cfg:
                emb.input_key = cfg["input_key"]
            elif "input_keys" in cfg:
                emb.input_keys = cfg["input_keys"]
            else:
                raise KeyError(f"need either 'input_key' or 'input_keys' for embedder {emb.__class__.__name__}")
            if emb.is_trainable:
                raise NotImplementedError("trainable embedders belong to the training path (out of scope)")
            for p in emb.parameters():
                p.requires_grad = False
            emb.eval()
            embedders.append(emb)
        self.embedders = nn.ModuleList(embedders)
        self.cor_embs, self.cor_p = list(cor_embs), list(cor_p)

    @torch.no_grad()
    def forward(self, batch: Dict, force_zero_embeddings: Optional[List[str]] = None) -> Dict:
        out: Dict[str, torch.Tensor] = {}
        force_zero_embeddings = force_zero_embeddings or []
        for emb in self.embedders:
            if getattr(emb, "input_key", None) is not None:
                res = emb(batch[emb.input_key])
            else:
                res = emb(*[batch[k] for k in emb.input_keys])
            if not isinstance(res, (torch.Tensor, list, tuple)):
                raise TypeError(f"encoder outputs must be tensors or a sequence, but got {type(res)}")
            for e in (res if isinstance(res, (list, tuple)) else [res]):
                key = self.OUTPUT_DIM2KEYS[e.dim()]
                if getattr(emb, "input_key", None) in force_zero_embeddings:
                    e = torch.zeros_like(e)
                out[key] = torch.cat((out[key], e), self.KEY2CATDIM[key]) if key in out else e
        return out

    def get_unconditional_conditioning(self, batch_c: Dict, batch_uc: Optional[Dict] = None,
                                       force_uc_zero_embeddings: Optional[List[str]] = None):
        c = self(batch_c)
        uc = self(batch_c if batch_uc is None else batch_uc, force_uc_zero_embeddings or [])
        return c, uc


def get_unique_embedder_keys_from_conditioner(conditioner: GeneralConditioner) -> List[str]:
    """sample_video.py:105-106."""
    return list({x.input_key for x in conditioner.embedders})


def get_batch(keys: Sequence[str], value_dict: Dict, N: Sequence[int], device="cuda"):
    """The ``txt`` branch of sample_video.py:109-178 (the only key SCAIL's conditioner has): the prompt and the negative
    prompt repeated over the batch; every other entry of ``value_dict`` is passed through; tensors are cloned into the
    unconditional batch."""
    n = 1
    for v in N:
        n *= int(v)
    batch, batch_uc = {}, {}
    for key in keys:
        if key == "txt":
            batch["txt"] = [value_dict["prompt"]] * n
            batch_uc["txt"] = [value_dict["negative_prompt"]] * n
        else:
            batch[key] = value_dict[key]
    for key, v in batch.items():
        if key not in batch_uc and isinstance(v, torch.Tensor):
            batch_uc[key] = v.clone()
    return batch, batch_uc

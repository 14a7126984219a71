"""Checkpoint ingest in the reference's on-disk format (sat/training/model_io.py:36-48, 233-327):
``<load>/latest`` holds the iteration (or ``release``); weights live in
``<load>/<iter>/mp_rank_{mp_rank:02d}_model_states.pt`` as ``{'module': state_dict, ...}`` whose DiT keys
are ``model.diffusion_model.<name>`` -- exactly the parameter paths of
``scail_amd.engine.SATVideoDiffusionEngine``, so the reference's files load unchanged.

Tensor-parallel checkpoints (``mp_rank_00 .. mp_rank_{N-1}``, written by a reference run with model_parallel_size = N):
this engine replicates the weights on every sequence-parallel rank, so the N partitions are MERGED
on load with the inverse of the reference's partition rule (sat/mpu/layers.py:286-340 ColumnParallelLinear.partition with
``stride``: the fused q|k|v and k|v projections are split per chunk; :420-500 RowParallelLinear: input dim, bias
replicated; driver sat/mpu/operation.py:96-124 mp_merge_model_rank0).  ``Wan2.1_VAE.pth`` is a bare state dict
(sgm/models/wan_vae.py:607-616)."""
from __future__ import annotations

import os
from typing import Tuple

import torch


def get_checkpoint_iteration(load_path: str) -> Tuple[int, bool]:
    tracker = os.path.join(load_path, "latest")
    if not os.path.isfile(tracker):
        raise ValueError(f"could not find the metadata file {tracker}, please check --load")
    meta = open(tracker).read().strip()
    try:
        return int(meta), False
    except ValueError:
        if meta != "release":
            raise ValueError(f"Invalid metadata file {tracker}")
        return 0, True


def get_checkpoint_name(load_path: str, iteration: int, release: bool = False, mp_rank: int = 0) -> str:
    d = "release" if release else f"{iteration:d}"
    return os.path.join(load_path, d, f"mp_rank_{mp_rank:02d}_model_states.pt")


# (suffix of the parameter's module path, kind, stride): the model-parallel layers on the DiT path
# (sat/model/transformer.py:56-176 SelfAttention / CrossAttention / MLP; dit...:985-1007 clip_feature_key_value_list)
_TP_RULES = (
    ("attention.query_key_value", "column", 3),
    ("cross_attention.query", "column", 1),
    ("cross_attention.key_value", "column", 2),
    ("mlp.dense_h_to_4h", "column", 1),
    ("attention.dense", "row", 1),            # matches cross_attention.dense as well
    ("mlp.dense_4h_to_h", "row", 1),
)


def _tp_rule(key: str):
    base, _, leaf = key.rpartition(".")
    if leaf not in ("weight", "bias"):
        return None, leaf
    if ".clip_feature_key_value_list." in key:
        return ("column", 2), leaf
    for suffix, kind, stride in _TP_RULES:
        if base.endswith(suffix):
            return (kind, stride), leaf
    return None, leaf


def merge_model_parallel_state_dicts(parts):
    """Merge the ``module`` state dicts of mp_rank_00 .. mp_rank_{N-1} into the full (model_parallel_size = 1) one.
    Column-parallel with stride s: rank r holds, for each of the s chunks (q | k | v), rows [r, r+1) x chunk/N -> the full
    weight is, chunk by chunk, the concatenation over ranks (layers.py:341-360 merge).  Row-parallel: concatenation along the
    input dim; its bias and every other parameter are replicated (checked equal)."""
    n = len(parts)
    if n == 1:
        return dict(parts[0])
    out = {}
    for key, t0 in parts[0].items():
        rule, leaf = _tp_rule(key)
        ts = [p[key] for p in parts]
        if rule is None or (rule[0] == "row" and leaf == "bias"):
            for r, t in enumerate(ts[1:], 1):
                if not torch.equal(t, t0):
                    raise ValueError(f"replicated parameter {key} differs between mp_rank_00 and mp_rank_{r:02d}")
            out[key] = t0
        elif rule[0] == "column":
            s = rule[1]
            if t0.shape[0] % s:
                raise ValueError(f"{key}: {t0.shape[0]} rows do not split into {s} strided chunks")
            per = t0.shape[0] // s
            out[key] = torch.cat([t[c * per:(c + 1) * per] for c in range(s) for t in ts], dim=0).contiguous()
        else:
            out[key] = torch.cat(ts, dim=1).contiguous()
    return out


def partition_state_dict(sd, n: int):
    """The reference's partition rule (layers.py:286-340, 438-460) applied to a full state dict: the ``module`` dicts a
    model_parallel_size = n run writes.  Used to produce reference-format multi-rank fixtures for the tests."""
    parts = [dict() for _ in range(n)]
    for key, t in sd.items():
        rule, leaf = _tp_rule(key)
        for r in range(n):
            if rule is None or (rule[0] == "row" and leaf == "bias"):
                parts[r][key] = t
            elif rule[0] == "column":
                s = rule[1]
                per = t.shape[0] // s
                sub = per // n
                parts[r][key] = torch.cat([t[c * per + r * sub:c * per + (r + 1) * sub] for c in range(s)], dim=0).contiguous()
            else:
                sub = t.shape[1] // n
                parts[r][key] = t[:, r * sub:(r + 1) * sub].contiguous()
    return parts


def model_parallel_files(load_path: str, iteration: int, release: bool):
    """mp_rank_XX files present for this iteration, in rank order."""
    names = []
    while os.path.isfile(get_checkpoint_name(load_path, iteration, release, len(names))):
        names.append(get_checkpoint_name(load_path, iteration, release, len(names)))
    return names


def load_checkpoint(module: torch.nn.Module, load_path: str, prefix: str = "", force_inference: bool = True, mp_rank: int = 0,
                    specific_iteration=None, merge_model_parallel: bool = True):
    """model_io.py:260-327 (inference branch): prefix filter, ``load_state_dict(strict=False)``; unexpected keys
    are reported, missing keys raise unless ``force_inference`` (yaml ``force_inference: True``).  When the directory holds
    several ``mp_rank_XX`` files they are merged (this engine keeps full weights on every rank)."""
    iteration, release = get_checkpoint_iteration(load_path)
    if specific_iteration is not None:
        iteration = int(specific_iteration)
    names = model_parallel_files(load_path, iteration, release) if merge_model_parallel else []
    if len(names) > 1:
        sd = {"module": merge_model_parallel_state_dicts([torch.load(nm, map_location="cpu")["module"] for nm in names])}
    else:
        name = get_checkpoint_name(load_path, iteration, release, mp_rank)
        sd = torch.load(name, map_location="cpu")
    mod = {k[len(prefix):]: v for k, v in sd["module"].items() if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(mod, strict=False)
    if unexpected:
        print(f"Will continue but found unexpected_keys! Check whether you are loading correct checkpoints: {unexpected}.")
    if missing:
        if not force_inference:
            raise ValueError(f"Missing keys for inference: {missing}.\nIf you still want to inference anyway, pass force_inference.")
        print(f"Warning: Missing keys for inference: {missing}.")
    return iteration, missing, unexpected


def save_checkpoint(module, save_path: str, iteration: int, mp_rank: int = 0, model_parallel_size: int = 1) -> str:
    """model_io.py:159-192 (module weights only), so round trips can be tested without the real files.  ``module`` is an
    nn.Module or a state dict; model_parallel_size = n writes the n partition files a tensor-parallel reference run would."""
    sd = module if isinstance(module, dict) else module.state_dict()
    parts = partition_state_dict(sd, model_parallel_size) if model_parallel_size > 1 else [sd]
    name = None
    for r, part in enumerate(parts):
        name = get_checkpoint_name(save_path, iteration, False, mp_rank + r)
        os.makedirs(os.path.dirname(name), exist_ok=True)
        torch.save({"module": part, "iteration": iteration}, name)
    with open(os.path.join(save_path, "latest"), "w") as f:
        f.write(str(iteration))
    return name


def load_vae_pth(model: torch.nn.Module, pretrained_path: str, device="cpu"):
    """sgm/models/wan_vae.py:607-616: ``Wan2.1_VAE.pth`` is a bare state dict loaded strictly into WanVAE_."""
    sd = torch.load(pretrained_path, map_location=device)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return model

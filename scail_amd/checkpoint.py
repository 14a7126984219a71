"""Checkpoint ingest in the reference's on-disk format (sat/training/model_io.py:36-48, 233-327):
``<load>/latest`` holds the iteration (or ``release``); weights live in
``<load>/<iter>/mp_rank_{mp_rank:02d}_model_states.pt`` as ``{'module': state_dict, ...}`` whose DiT keys
are ``model.diffusion_model.<name>`` -- exactly the parameter paths of
``scail_amd.engine.SATVideoDiffusionEngine``, so the reference's files load unchanged."""
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


def load_checkpoint(module: torch.nn.Module, load_path: str, prefix: str = "", force_inference: bool = True, mp_rank: int = 0,
                    specific_iteration=None):
    """model_io.py:260-327 (inference branch): prefix filter, ``load_state_dict(strict=False)``; unexpected keys
    are reported, missing keys raise unless ``force_inference`` (yaml ``force_inference: True``)."""
    iteration, release = get_checkpoint_iteration(load_path)
    if specific_iteration is not None:
        iteration = int(specific_iteration)
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


def save_checkpoint(module: torch.nn.Module, save_path: str, iteration: int, mp_rank: int = 0) -> str:
    """model_io.py:159-192 (module weights only), so round trips can be tested without the real files."""
    name = get_checkpoint_name(save_path, iteration, False, mp_rank)
    os.makedirs(os.path.dirname(name), exist_ok=True)
    torch.save({"module": module.state_dict(), "iteration": iteration}, name)
    with open(os.path.join(save_path, "latest"), "w") as f:
        f.write(str(iteration))
    return name

"""HF-checkpoint ingest: the on-disk layout written by the reference's converter
(Pai-Megatron-Patch/toolkits/model_checkpoints_convertor/llava/hf2megatron_llava.py:1050-1484):

    <dir>/config.json                      LlavaConfig (vision_config, text_config, image_token_index=128200, ...)
    <dir>/model.safetensors | model-0000x-of-0000y.safetensors + model.safetensors.index.json
    <dir>/pytorch_model.bin  | pytorch_model-0000x-of-0000y.bin + pytorch_model.bin.index.json
    keys: language_model.model.*, language_model.lm_head.weight, multi_modal_projector.linear_{1,2}.*,
          vision_tower.vision_model.*   (the SigLIP pooling head `vision_tower.vision_model.head.*` is ignored:
          the reference never uses its output, EVAL:273)

``CheckpointSource`` streams one tensor at a time to the device (fp32 on disk -> 16-bit compute type for matrices,
fp32 for vectors), so an 8B-parameter fp32 checkpoint never needs to be resident in host RAM.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Optional

import torch

from .config import LeopardConfig
from .synth import param_specs


class CheckpointSource:
    def __init__(self, path: str, device, dtype):
        self.path, self.device, self.dtype = path, device, dtype
        self._where: Dict[str, str] = {}
        self._bin_cache: Dict[str, dict] = {}
        st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        bins = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
        if st:
            from safetensors import safe_open
            for f in st:
                with safe_open(f, framework="pt") as h:
                    for k in h.keys():
                        self._where[k] = f
        elif bins:
            for f in bins:
                sd = torch.load(f, map_location="cpu", weights_only=True, mmap=True)
                self._bin_cache[f] = sd
                for k in sd:
                    self._where[k] = f
        else:
            raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {path}")

    def keys(self):
        return self._where.keys()

    def get(self, name: str) -> torch.Tensor:
        f = self._where.get(name)
        if f is None:
            raise KeyError(f"checkpoint {self.path} has no tensor {name!r}")
        if f in self._bin_cache:
            t = self._bin_cache[f][name]
        else:
            from safetensors import safe_open
            with safe_open(f, framework="pt") as h:
                t = h.get_tensor(name)
        t = t.to(device=self.device, dtype=self.dtype if t.dim() >= 2 else torch.float32)
        if t.data_ptr() % 16:                      # views into a memory-mapped file can sit at any 4-byte offset
            t = t.clone()
        return t


def load_config(path: str) -> LeopardConfig:
    return LeopardConfig.load(os.path.join(path, "config.json"))


def save_synthetic_checkpoint(path: str, cfg: LeopardConfig, shard_bytes: Optional[int] = None) -> None:
    """Write a checkpoint directory in the converter's layout from the seeded synthetic parameters (fp32
    safetensors, optionally sharded with an index file).  Used by tests and to smoke the eval harness: no released
    weights exist offline."""
    from safetensors.torch import save_file
    from .synth import synth_array
    os.makedirs(path, exist_ok=True)
    cfg.save(os.path.join(path, "config.json"))
    tensors = {n: torch.from_numpy(synth_array(n, s, k)) for n, s, k in param_specs(cfg)}
    if not shard_bytes:
        save_file(tensors, os.path.join(path, "model.safetensors"))
        return
    shards, cur, size = [], {}, 0
    for n, t in tensors.items():
        b = t.numel() * 4
        if cur and size + b > shard_bytes:
            shards.append(cur)
            cur, size = {}, 0
        cur[n] = t
        size += b
    shards.append(cur)
    index = {"metadata": {}, "weight_map": {}}
    for i, sh in enumerate(shards):
        fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(path, fn))
        for n in sh:
            index["weight_map"][n] = fn
    with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
        json.dump(index, f)

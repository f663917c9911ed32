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
        # what the cast of the matrices to the 16-bit compute type cost (round 6): the released Leopard checkpoints are
        # bf16-trained (train_multiimg_llava_siglip.sh:64), and a bf16 value is exact in fp16 down to |w| = 2^-17 (fp16 subnormals keep 8
        # significant bits that far) — below that it loses bits, below 2^-25 it flushes to zero.  An fp32-VALUED checkpoint is rounded
        # (relative 2^-12 per weight), which no precision mode corrects: the loader counts both cases so that the caller can say so.
        self.cast_stats = {"matrices": 0, "inexact_matrices": 0, "inexact_elements": 0, "flushed_to_zero": 0, "max_abs_error": 0.0, "elements": 0}
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
        src = t
        t = t.to(device=self.device, dtype=self.dtype if t.dim() >= 2 else torch.float32)
        if src.dim() >= 2 and src.dtype != t.dtype:
            back = src.to(self.device, torch.float32) - t.to(torch.float32)
            bad = back != 0
            st = self.cast_stats
            st["matrices"] += 1
            st["elements"] += src.numel()
            n_bad = int(bad.sum())
            if n_bad:
                st["inexact_matrices"] += 1
                st["inexact_elements"] += n_bad
                st["flushed_to_zero"] += int(((t == 0) & bad).sum())
                st["max_abs_error"] = max(st["max_abs_error"], float(back.abs().max()))
        if t.data_ptr() % 16:                      # views into a memory-mapped file can sit at any 4-byte offset
            t = t.clone()
        return t

    def cast_report(self) -> str:
        st = self.cast_stats
        if not st["inexact_elements"]:
            return f"{st['matrices']} matrices cast to {str(self.dtype).replace('torch.', '')} exactly"
        return (f"{st['inexact_elements']} of {st['elements']} matrix elements ({st['inexact_matrices']} of {st['matrices']} matrices) are NOT exactly "
                f"representable in {str(self.dtype).replace('torch.', '')}: max |error| {st['max_abs_error']:.3e}, {st['flushed_to_zero']} flushed to zero — "
                "the weight cast is a rounding that no precision mode corrects")


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

"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / gloo on CPU.

How the path shards:
  * by SAMPLE — the reference's own scheme (run_eval_llava_siglip_multiimg.sh:9-11 + eval_utils.split_shard): replicas,
    no data-path collective.  ``shard_records`` is that split; bench.py uses ``barrier`` / ``max_over_ranks`` only.
  * by ViT INPUT (tile) inside one sample: the 676-token tile sequences are independent through the tower and the
    projector, so rank r encodes a contiguous, balanced slice of the tiles and ONE all-gather of the projected visual
    tokens [N_r*169, 4096] restores the full, correctly ordered set on every rank (``encode_images_sharded``).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> Tuple[int, int]:
    """Initialise the default process group from the torchrun environment.  Returns (rank, world).
    backend None = RCCL ("nccl") when a GPU is present, with gloo as the fallback if RCCL cannot come up: the data path has
    no collective (sample sharding), so the group only carries the barrier and the max-over-ranks timing reduce."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        want = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if want == "nccl":
            try:
                kw = {"device_id": device} if device is not None else {}
                dist.init_process_group("nccl", rank=rank, world_size=world, **kw)
                probe = torch.zeros(1, device=device if device is not None else "cuda")
                dist.all_reduce(probe)                              # force communicator creation now, not in the timed region
                torch.cuda.synchronize()
            except Exception as e:                                  # pragma: no cover  (needs a multi-GPU node)
                print(f"[leopard_amd.dist] RCCL init failed on rank {rank} ({e!r}); falling back to gloo", flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(want, rank=rank, world_size=world)
    return rank, world


def _reduce_device(device):
    """gloo reduces host tensors; RCCL reduces device tensors."""
    return "cpu" if dist.get_backend() == "gloo" else device


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_records(records: list, rank: int, world: int) -> list:
    """eval_utils.split_shard (evaluations/models/eval_utils.py:84-89): contiguous slices of len//world + 1."""
    size = len(records) // world + 1
    return records[rank * size:(rank + 1) * size]


def tile_slices(n_tiles: int, world: int) -> List[Tuple[int, int]]:
    """Balanced contiguous tile ranges: the first n_tiles % world ranks take one extra tile."""
    base, extra = divmod(n_tiles, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def encode_images_sharded(engine, tiles: torch.Tensor) -> torch.Tensor:
    """Vision tower + projector with the tiles split across the ranks of the default group, then ONE all-gather of
    the projected visual tokens.  Every rank returns the full fp32 [N*tokens_per_tile, D] tensor in tile order —
    bit-identical to ``engine.encode_images(tiles)`` because tiles are independent sequences."""
    world = world_size()
    if world == 1:
        return engine.encode_images(tiles)
    rank = dist.get_rank()
    slices = tile_slices(tiles.shape[0], world)
    lo, hi = slices[rank]
    tpt = engine.cfg.tokens_per_tile
    D = engine.cfg.text_config.hidden_size
    max_rows = max(b - a for a, b in slices) * tpt
    mine = torch.zeros(max_rows, D, dtype=torch.float32, device=tiles.device)
    if hi > lo:
        mine[:(hi - lo) * tpt] = engine.encode_images(tiles[lo:hi].contiguous())
    gathered = torch.empty(world * max_rows, D, dtype=torch.float32, device=tiles.device)
    if dist.get_backend() == "gloo" and tiles.device.type == "cuda":            # fallback group: stage through the host
        host = torch.empty(world * max_rows, D, dtype=torch.float32)
        dist.all_gather_into_tensor(host, mine.cpu())
        gathered.copy_(host)
    else:
        dist.all_gather_into_tensor(gathered, mine)              # equal-size padded shards, one collective
    parts = [gathered[r * max_rows:r * max_rows + (b - a) * tpt] for r, (a, b) in enumerate(slices)]
    return torch.cat(parts, dim=0)


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks (the tensor-parallel exchange of the LLM: partial o_proj / down_proj products).  RCCL reduces
    the device tensor on the current stream; gloo (CPU tests) reduces host tensors."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and t.device.type == "cuda":      # fallback group: stage through the host
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


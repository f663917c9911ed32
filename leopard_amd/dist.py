"""Multi-GPU plumbing: one process per GPU.

How the path shards:
  * by SAMPLE — the reference's own scheme (run_eval_llava_siglip_multiimg.sh:9-11 + eval_utils.split_shard): replicas,
    no data-path collective.  ``shard_records`` is that split; bench.py uses ``barrier`` / ``max_over_ranks`` only.
  * ONE sample on all ranks (north_star: "shard the per-image ViT encodes and LLM tensor-parallel across the 8 GPUs"):
      - the 676-token ViT inputs are independent through the tower and the projector: rank r encodes a balanced contiguous
        slice of them and ONE all-gather of the projected visual tokens restores the full set (``encode_images_sharded``);
      - the LLM runs tensor-parallel with SEQUENCE-PARALLEL norms (the Megatron exchange pattern,
        Megatron-LM-240603/megatron/core/tensor_parallel/mappings.py:107-145, layers.py:387-454): the fp32 residual stream is
        sharded by rows; per half layer  RMSNorm(local rows) -> all-gather (16-bit) -> column-parallel GEMM -> ... -> row-parallel
        GEMM -> reduce-scatter (16-bit) -> residual add on the local rows  (LeopardEngine._llm_prefill_tp).

Transports (``Comm``): on GPUs the data path uses RCCL through the C ABI (``RcclComm`` over lmi_comm_init / lmi_allgather /
lmi_reduce_scatter / lmi_allreduce, stream-ordered, so collectives can run on a side stream under the next GEMM); the
128-byte RCCL unique id travels over torch.distributed (backend "nccl" = RCCL), which also carries the barriers of bench.py.
``TorchComm`` drives the same algorithms over any torch.distributed group — gloo in the CPU tests (emulated kernels) and in the
functional 2-processes-on-one-GPU test.  There is NO path from a GPU job to gloo: if RCCL cannot come up, ``init`` raises; if only
the C-ABI communicator cannot (agreed on by all ranks), the same algorithm runs over torch.distributed's RCCL group and the
transport name says so (LMI_COMM_STRICT=1 makes that an error too).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import warnings
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib


def init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> Tuple[int, int]:
    """Initialise the default process group from the torchrun environment.  Returns (rank, world).
    backend None = RCCL ("nccl") when a GPU is present, gloo otherwise.  A failing RCCL initialisation is an error (a job that
    reports N ranks must have N RCCL ranks); pass backend="gloo" explicitly for CPU runs / the one-GPU functional test."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        want = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if want == "nccl":
            # LMI_TP_COMM_CUS: upper bound on the workgroups (CUs) RCCL's transport kernels take — RCCL's own knob NCCL_MAX_NCHANNELS,
            # set before any communicator exists.  The tensor-parallel prefill runs its exchanges on a side stream UNDER GEMMs that hold
            # one 128 KiB-LDS workgroup on every CU; tools/overlap_probe.py (profiles/r03_overlap_probe.txt) shows that a transport-like
            # kernel of <= 64 workgroups is co-scheduled beside them at full rate and what it costs the GEMMs, which is where the
            # default comes from.  Unset = RCCL's own choice.
            cus = os.environ.get("LMI_TP_COMM_CUS")
            if cus:
                os.environ.setdefault("NCCL_MAX_NCHANNELS", str(int(cus)))
            kw = {"device_id": device} if device is not None else {}
            dist.init_process_group("nccl", rank=rank, world_size=world, **kw)
            probe = torch.ones(1, device=device if device is not None else "cuda")
            dist.all_reduce(probe)                                  # force communicator creation now, not in the timed region
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"RCCL all-reduce over {world} ranks returned {probe.item()}")
        else:
            dist.init_process_group(want, rank=rank, world_size=world)
    return rank, world


def backend_name() -> str:
    return dist.get_backend() if dist.is_initialized() else "none"


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


# ------------------------------------------------------------------------------------------------------------------------------
# transports
# ------------------------------------------------------------------------------------------------------------------------------
class Comm:
    """Collectives of the one-sample-on-all-ranks path.  ``stream`` (a torch.cuda.Stream or None = current) is where the
    collective is enqueued; ``sent_bytes`` accumulates the bytes this rank puts on its links, counted for the direct (full-mesh)
    algorithms xGMI favours: all-gather (R-1) x shard, reduce-scatter (R-1)/R x input, all-reduce 2 (R-1)/R x tensor."""
    rank: int = 0
    world: int = 1
    backend: str = "none"
    sent_bytes: int = 0

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor, stream=None): raise NotImplementedError
    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor, stream=None): raise NotImplementedError
    def all_reduce(self, t: torch.Tensor, stream=None): raise NotImplementedError
    def broadcast(self, t: torch.Tensor, root: int, stream=None): raise NotImplementedError
    def ranks_seen(self) -> int: return 0                       # ranks RCCL itself reports (0: not an RCCL transport)

    def _count(self, kind: str, t: torch.Tensor):
        b, R = t.numel() * t.element_size(), self.world
        self.sent_bytes += {"ag": (R - 1) * b, "rs": (R - 1) * b // R, "ar": 2 * (R - 1) * b // R}[kind]


class TorchComm(Comm):
    """torch.distributed default group (gloo: CPU tests and the two-processes-on-one-GPU functional test; also usable with
    "nccl").  A gloo group reduces host tensors, so device tensors are staged through the host — functional checks only."""

    def __init__(self):
        assert dist.is_initialized()
        self.rank, self.world, self.backend = dist.get_rank(), dist.get_world_size(), dist.get_backend()
        self._group_backend = self.backend                     # ``backend`` is a display name (get_comm may extend it)
        self.sent_bytes = 0

    def _host_staged(self, t):
        return self._group_backend == "gloo" and t.device.type == "cuda"

    def ranks_seen(self) -> int:
        return self.world if self._group_backend == "nccl" else 0

    @staticmethod
    def _on(stream):
        """torch.distributed orders a collective against the CURRENT stream: make ``stream`` current, so that the side-stream
        overlap of the engine holds for this transport as it does for RcclComm."""
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def all_gather(self, out, inp, stream=None):
        self._count("ag", inp)
        with self._on(stream):
            if self._host_staged(inp):
                h = torch.empty(out.shape, dtype=out.dtype)
                dist.all_gather_into_tensor(h, inp.cpu().contiguous())
                out.copy_(h)
            else:
                dist.all_gather_into_tensor(out, inp)

    def reduce_scatter(self, out, inp, stream=None):
        self._count("rs", inp)
        with self._on(stream):
            if self._host_staged(inp):
                h = torch.empty(out.shape, dtype=out.dtype)
                dist.reduce_scatter_tensor(h, inp.cpu().contiguous())
                out.copy_(h)
            else:
                dist.reduce_scatter_tensor(out, inp)

    def all_reduce(self, t, stream=None):
        self._count("ar", t)
        with self._on(stream):
            if self._host_staged(t):
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def broadcast(self, t, root, stream=None):
        with self._on(stream):
            if self._host_staged(t):
                h = t.cpu()
                dist.broadcast(h, src=root)
                t.copy_(h)
            else:
                dist.broadcast(t, src=root)


_LMI_DT = {torch.float16: _lib.LMI_F16, torch.bfloat16: _lib.LMI_BF16, torch.float32: _lib.LMI_F32}


class RcclComm(Comm):
    """RCCL communicator owned by libleopard_amd.so (lmi_comm_init): collectives are plain stream-ordered C-ABI calls on raw
    device pointers.  The unique id is created by rank 0 and broadcast over the torch.distributed group (the bootstrap)."""
    backend = "rccl (lmi_comm)"

    def __init__(self, lib=None, device: Optional[torch.device] = None):
        assert dist.is_initialized(), "bootstrap group missing: call leopard_amd.dist.init() first"
        self.lib = lib if lib is not None else _lib.load()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.sent_bytes = 0
        # Rank 0 creates the unique id and EVERY rank takes part in its broadcast, whatever happened on rank 0: the message is one
        # status byte + the 128 id bytes.  (Raising on rank 0 before the broadcast would leave the other ranks blocked in it while
        # rank 0 moved on to the next collective — mismatched collectives, a hang instead of the agreed fallback in get_comm.)
        uid, status, err = (C.c_char * 128)(), 1, ""
        if self.rank == 0:
            rc = self.lib.lmi_comm_unique_id(uid)
            if rc != 0:
                status, err = 0, f"libleopard_amd comm error {rc}: {self.lib.lmi_last_error().decode()}"
        t = torch.tensor([status] + list(bytes(uid)), dtype=torch.uint8)
        t = t.to(self.device) if dist.get_backend() != "gloo" else t
        dist.broadcast(t, src=0)
        msg = t.cpu().tolist()
        if msg[0] != 1:
            raise RuntimeError(err or "rank 0 could not create the RCCL unique id (lmi_comm_unique_id)")
        uid = (C.c_char * 128).from_buffer_copy(bytes(msg[1:]))
        handle = C.c_void_p()
        torch.cuda.set_device(self.device)
        self._check(self.lib.lmi_comm_init(self.rank, self.world, uid, C.byref(handle)))
        self.handle = handle

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"libleopard_amd comm error {rc}: {self.lib.lmi_last_error().decode()}")

    def _s(self, stream):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        return C.c_void_p(s.cuda_stream)

    def ranks_seen(self) -> int:
        return int(self.lib.lmi_comm_size(self.handle))

    def all_gather(self, out, inp, stream=None):
        assert inp.is_contiguous() and out.is_contiguous() and out.numel() == inp.numel() * self.world and out.dtype == inp.dtype
        self._count("ag", inp)
        self._check(self.lib.lmi_allgather(self.handle, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()), inp.numel(),
                                           _LMI_DT[inp.dtype], self._s(stream)))

    def reduce_scatter(self, out, inp, stream=None):
        assert inp.is_contiguous() and out.is_contiguous() and inp.numel() == out.numel() * self.world and out.dtype == inp.dtype
        self._count("rs", inp)
        self._check(self.lib.lmi_reduce_scatter(self.handle, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()), out.numel(),
                                                _LMI_DT[inp.dtype], self._s(stream)))

    def all_reduce(self, t, stream=None):
        assert t.is_contiguous()
        self._count("ar", t)
        self._check(self.lib.lmi_allreduce(self.handle, C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), t.numel(), _LMI_DT[t.dtype],
                                           self._s(stream)))

    def broadcast(self, t, root, stream=None):
        assert t.is_contiguous()
        self._check(self.lib.lmi_broadcast(self.handle, C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), t.numel(), _LMI_DT[t.dtype],
                                           int(root), self._s(stream)))

    def destroy(self):
        if getattr(self, "handle", None):
            self.lib.lmi_comm_destroy(self.handle)
            self.handle = None


_COMM: Optional[Comm] = None


def get_comm(device=None, lib=None) -> Optional[Comm]:
    """The communicator of the one-sample-on-all-ranks path: RCCL through the C ABI when the bootstrap group is "nccl",
    the torch.distributed group otherwise (gloo: CPU / functional tests).  None on a single rank."""
    global _COMM
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    if _COMM is None:
        if dist.get_backend() != "nccl":
            _COMM = TorchComm()
        else:
            # RCCL through the C ABI.  Whether it came up is agreed on by ALL ranks (a rank that failed must not leave the others
            # inside ncclCommInitRank's rendezvous with a different transport).  If it did not, the SAME algorithm runs over the
            # torch.distributed group — which is RCCL as well — and says so in ``backend`` (bench.py prints it); LMI_COMM_STRICT=1
            # turns that into an error.  There is no path from a GPU job to gloo.
            comm, err = None, ""
            try:
                comm = RcclComm(lib=lib, device=device)
            except Exception as e:                                   # noqa: BLE001  (re-raised or reported below)
                err = repr(e)[:300]
            flag_dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu")
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm is not None:
                    comm.destroy()
                if os.environ.get("LMI_COMM_STRICT") == "1":
                    raise RuntimeError(f"RCCL communicator through the C ABI unavailable on at least one rank: {err or 'other rank'}")
                warnings.warn(f"lmi_comm_init failed on at least one rank ({err or 'other rank'}): collectives run over torch.distributed's RCCL group")
                comm = TorchComm()
                comm.backend = f"rccl (torch.distributed nccl; lmi_comm unavailable: {err or 'failed on another rank'})"
            _COMM = comm
    return _COMM


def reset_comm():
    global _COMM
    if isinstance(_COMM, RcclComm):
        _COMM.destroy()
    _COMM = None


def encode_images_sharded(engine, tiles: torch.Tensor) -> torch.Tensor:
    """Vision tower + projector with the ViT inputs split across the ranks, then ONE all-gather of the projected visual
    tokens.  Every rank returns the full fp32 [N*tokens_per_tile, D] tensor in tile order — bit-identical to
    ``engine.encode_images(tiles)`` because tiles are independent sequences."""
    comm = get_comm(tiles.device if tiles.is_cuda else None, getattr(engine.ops, "lib", None)) if engine.comm is None else engine.comm
    if comm is None:
        return engine.encode_images(tiles)
    world, rank = comm.world, comm.rank
    slices = tile_slices(tiles.shape[0], world)
    lo, hi = slices[rank]
    tpt = engine.cfg.tokens_per_tile
    D = engine.cfg.text_config.hidden_size
    max_rows = max(b - a for a, b in slices) * tpt
    # exchange dtype: fp32 by default — the gathered rows are merged into the fp32 residual stream, which carries them UNROUNDED through
    # every layer, so a 16-bit exchange is one rounding the single-rank path does not have (2^-11 relative in fp16, 2^-8 in bf16) and the
    # tensor-parallel result would no longer equal the one-rank result.  116 MB at C3 over 7 links is ~0.1 ms.  `engine.tp_vision_gather_dtype
    # = engine.dtype` opts into the 16-bit exchange (58 MB, the figure SURVEY.md 8e budgets) as a bandwidth mode with that stated rounding.
    gdt = getattr(engine, "tp_vision_gather_dtype", None) or torch.float32
    mine = torch.zeros(max_rows, D, dtype=gdt, device=tiles.device)
    if hi > lo:
        mine[:(hi - lo) * tpt] = engine.encode_images(tiles[lo:hi].contiguous())
    gathered = torch.empty(world * max_rows, D, dtype=gdt, device=tiles.device)
    comm.all_gather(gathered, mine)                               # equal-size padded shards, one collective
    parts = [gathered[r * max_rows:r * max_rows + (b - a) * tpt] for r, (a, b) in enumerate(slices)]
    return torch.cat(parts, dim=0).float()

"""Prefill / decode orchestration of the Leopard-LLaVA hot path on one MI355X.

Host Python only sequences kernel launches on the current HIP stream; every arithmetic step is a call into
libleopard_amd.so (see leopard_amd/ops.py).  The data path mirrors the reference forward
(evaluations/models/llava_multiimg_siglip_anyres.py:261-333, "EVAL"):

    u8 tiles / pixel_values --lmi_preprocess_tiles--> im2col rows
      --GEMM(+bias +pos-emb)--> fp32 ViT stream --27x[LN, QKV GEMM, varlen attention (one 676-token sequence per
      tile), out-proj GEMM(+residual), LN, fc1 GEMM(+gelu_tanh), fc2 GEMM(+residual)]--> post-LN           (EVAL:268-273)
      --GEMM(pixel-shuffle gather, +gelu_erf)--GEMM--> visual tokens [N*169, 4096] fp32                      (EVAL:283)
      --lmi_embed_merge (host-planned index map)--> fp32 LLM stream [S, 4096]                                (EVAL:263,285)
      --32x[RMSNorm, QKV GEMM, RoPE(+KV cache), causal GQA attention, o GEMM(+residual), RMSNorm,
            gate/up GEMM(+SwiGLU), down GEMM(+residual)]--> final RMSNorm --> lm_head                        (EVAL:322-333)

HBM residency: the residual streams are fp32 (ViT [N*676,1152], LLM [S,4096]); GEMM operands and activations
between kernels are the 16-bit compute type; weights are converted once at load (leopard_amd/weights.py).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .config import LeopardConfig
from .ops import Ops
from .weights import EngineWeights


def llama3_inv_freq(head_dim: int, theta: float, scaling) -> torch.Tensor:
    """Inverse RoPE frequencies with the llama3.1 wavelength-dependent scaling
    (Megatron-LM-240603/megatron/core/models/common/embeddings/rotary_pos_embedding.py:48-83)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    if scaling is None:
        return inv
    lo_wl = scaling.original_max_position_embeddings / scaling.low_freq_factor
    hi_wl = scaling.original_max_position_embeddings / scaling.high_freq_factor
    wl = 2 * math.pi / inv
    scaled = torch.where(wl > lo_wl, inv / scaling.factor, inv)
    smooth = (scaling.original_max_position_embeddings / wl - scaling.low_freq_factor) / (
        scaling.high_freq_factor - scaling.low_freq_factor)
    mid = (1 - smooth) * scaled / scaling.factor + smooth * scaled
    return torch.where(~(wl < hi_wl) & ~(wl > lo_wl), mid, scaled)


def plan_merge(input_ids: np.ndarray, image_token_index: int, n_feature_rows: int, tokens_per_tile: int) -> np.ndarray:
    """Index map of the merged sequence (transformers-4.38 ``_merge_input_ids_with_image_features`` semantics for
    one unpadded sample, EVAL:284-287): ``src[s] >= 0`` -> text embedding of input position ``src[s]``;
    ``src[s] < 0`` -> visual token row ``-src[s]-1``.  Raises ValueError on an image-token / feature count
    mismatch, before any kernel is launched."""
    ids = np.asarray(input_ids, dtype=np.int64).reshape(-1)
    is_img = ids == image_token_index
    n_img = int(is_img.sum())
    if n_img * tokens_per_tile != n_feature_rows:
        raise ValueError(
            f"The input provided to the model are wrong. The number of image tokens is {n_img} while the number of "
            f"image given to the model is {n_feature_rows // max(tokens_per_tile, 1)}. This prevents correct indexing "
            "and breaks batch generation.")
    width = np.where(is_img, tokens_per_tile, 1)
    start = np.cumsum(width) - width
    S = int(width.sum())
    src = np.empty(S, dtype=np.int64)
    text_pos = np.nonzero(~is_img)[0]
    src[start[text_pos]] = text_pos
    img_pos = np.nonzero(is_img)[0]
    if n_img:
        rows = (start[img_pos][:, None] + np.arange(tokens_per_tile)[None, :]).reshape(-1)
        src[rows] = -(np.arange(n_img * tokens_per_tile) + 1)
    return src


@dataclass
class PrefillResult:
    logits_last: torch.Tensor                    # fp32 [vocab]
    seq_len: int
    n_tiles: int
    logits_all: Optional[torch.Tensor] = None    # fp32 [S, vocab] when requested (what the reference computes)
    parts: Optional[Dict[str, torch.Tensor]] = None


class KVCache:
    def __init__(self, cfg: LeopardConfig, capacity: int, dtype, device, tp_size: int = 1):
        tc = cfg.text_config
        w = tc.num_key_value_heads // tp_size * tc.head_dim          # a tensor-parallel rank caches its own kv heads
        self.k = [torch.zeros(capacity, w, dtype=dtype, device=device) for _ in range(tc.num_hidden_layers)]
        self.v = [torch.zeros(capacity, w, dtype=dtype, device=device) for _ in range(tc.num_hidden_layers)]
        self.capacity, self.length = capacity, 0


class LeopardEngine:
    def __init__(self, cfg: LeopardConfig, weights: EngineWeights, ops: Optional[Ops] = None, device=None,
                 use_tr: bool = True, comm=None, pack_llm_weights: Optional[bool] = None):
        self.cfg, self.W = cfg, weights
        self.ops = ops if ops is not None else Ops()
        self.dtype = weights.dtype
        self.device = torch.device(device) if device is not None else weights.embed.device
        self.use_tr = use_tr
        self.use_graphs = True         # capture the decode step in a HIP graph (cuda devices only)
        # tensor parallel (weights built with tp_size > 1): the communicator of the one-sample-on-all-ranks path (leopard_amd.dist)
        self.comm = comm
        if getattr(weights, "tp_size", 1) > 1 and comm is None:
            from . import dist as D
            self.comm = D.get_comm(self.device if self.device.type == "cuda" else None, getattr(self.ops, "lib", None))
            if self.comm is None or self.comm.world != weights.tp_size:
                raise RuntimeError(f"tensor-parallel weights (tp_size {weights.tp_size}) need an initialised process group of that size")
        self.tp_chunks = 2             # row chunks per layer under TP: chunk c's collectives overlap chunk c+1's GEMMs
        # dtype of the reduce-scattered partial products.  torch.float32 (default since round 5) = exact partial sums: at C2 full depth two ranks
        # with fp32 sums sit at 1.18e-3 of the logit scale against the fp32 reference — the one-rank fast schedule's 1.14e-3 — while the
        # 16-bit exchange (None = the compute type: half the reduce-scatter bytes) adds a rounding per partial product and half layer, 1.74e-3
        # (tests/test_gpu_dist.py, profiles/r05_tp_parity.txt).  LMI_TP_COMM_DTYPE=16 selects the 16-bit exchange.
        self.tp_comm_dtype = None if os.environ.get("LMI_TP_COMM_DTYPE", "32") == "16" else torch.float32
        # TP decode: capture the step (with its RCCL all-reduces) in a HIP graph when the communicator is RcclComm.  OFF by default: capture and
        # replay of a multi-rank RCCL step has only ever run on a one-rank communicator (tests/test_gpu_dist.py) — LMI_TP_DECODE_GRAPH=1 /
        # this flag opt in, and a failed capture falls back to the eager step (``_decode_run``)
        self.tp_decode_graph = os.environ.get("LMI_TP_DECODE_GRAPH", "0") == "1"
        # all-gather of the projected visual tokens: None = fp32 (bit-identical to one rank: the rows are merged into the fp32 residual stream,
        # which carries them unrounded through every layer); the 16-bit compute type halves the bytes (58 MB at C3) at one extra rounding
        self._comm_stream = None
        self._workspaces: Dict[tuple, torch.Tensor] = {}   # caller-owned scratch per (stage, launch stream) (lmi_llm_prefill_workspace_bytes / lmi_vit_workspace_bytes)
        self.graph_encode = False      # capture the vision encode per ViT-input count in a HIP graph (BASELINE config 5)
        self._encode_graphs: Dict[tuple, tuple] = {}   # (ViT-input count, stream) -> (graph, static in, static out)
        self._private_scratch = False  # True while an encode graph is warmed up / captured: _carve hands out fresh allocations the graph owns
        self.fuse_norm_rope = True     # Llama layers: RMSNorm + RoPE + KV append inside the GEMM epilogues (lmi_rmsnorm_rope / lmi_gemm_ex)
        self.suppress_tokens = None    # optional int64 device tensor of token ids that greedy decoding may never emit (HF bad_words_ids)
        self.trace = None              # optional callable(name, fp32 residual stream) after the embeddings / every layer (tests)
        # Split-operand precision mode: every A operand of every ViT / LLM layer linear is handed over as a hi + lo pair of
        # 16-bit values (lmi_split_hi_lo) and multiplied against [W | W] — the GEMMs run at 2 K, the hand-over roundings that make up the
        # distance to the fp32 reference are gone (full-depth logits within north_star's 1e-3; ~1.8x the prefill time).  Prefill only.
        self.split_operands = False
        # Low-bit correction mode (round 5): the same goal at + 25 % matrix time instead of + 100 %.  Every producer of a layer-
        # linear A operand also hands over an MX fp4 image of the rounding residual x - T(x) (lmi_norm_lo4, lmi_split_lo4, the GELU / SwiGLU /
        # folded-norm GEMM epilogues), every layer linear has an fp4 weight image (+ 0.5 B per parameter), and the GEMMs run a second k-loop
        # phase of v_mfma_scale_f32_32x32x64_f8f6f4 on the two images into the accumulators of the 16-bit pass (lmi_gemm_lo4).  Prefill only;
        # one rank.  ``precision`` = "fast" | "lo4" | "split" selects between the three schedules.
        self.lo4 = False
        # lo4 corrects the LLM layer linears; ``lo4_vit`` (LMI_LO4_VIT=1) extends it to the SigLIP layer linears.  Off by default: measured at
        # full depth, the tower's correction moves the logits of the benchmarked C3 sample by < 1 % (2.35e-4 vs 2.37e-4 of the logit scale) and
        # those of the hardest case — C1: one ViT input, S = 228 — from 7.3e-4 to 6.1e-4, for + 6 % of the step (1.20 x vs 1.27 x the fast schedule)
        # Round 6: "auto" (default) = the tower is corrected too for the samples whose LLM sequence is short (<= LO4_FULL_BELOW rows: C1-like, where
        # the rounding noise of ONE realisation lands anywhere between 7e-4 and 1e-3 without it) — a per-SAMPLE decision carried to the tower's rows
        # by the same row selection as the LLM's (the tiles of a long sample in a packed batch stay on the fast tower: packed == separate).
        env_vit = os.environ.get("LMI_LO4_VIT", "auto")
        self.lo4_vit = "auto" if env_vit == "auto" else env_vit == "1"
        self._lo4_w = {}               # tower -> fp4 weight images (built when the mode is selected / on first use; dropped by invalidate_lo4_weights)
        # WHICH ROWS carry the correction (round 6; DESIGN.md 2.1 "row selection", tools/lo4_policy_study.py).  The logits of a row are dominated by
        # the hand-over roundings on that row's OWN path through the 32 layers; the roundings of the other rows reach it only through the softmax
        # average over the keys, i.e. attenuated by ~sqrt(S) (emulating oracle, C1: correcting ONLY the last row removes 90 % of what correcting all
        # 228 rows removes).  The rows whose logits are read are the last rows of each sequence, so by default ("auto") a sequence longer than
        # LO4_FULL_BELOW rows carries the correction on its last LO4_TAIL_ROWS rows only; shorter sequences on every row.  "all" = every row (round
        # 5's schedule); an int = that many trailing rows.  A property of the ROW (its distance from the end of its sequence), not of where the row
        # lands in a tile: packed == separate stays bit for bit.  LMI_LO4_ROWS overrides.
        env_rows = os.environ.get("LMI_LO4_ROWS", "auto")
        self.lo4_rows = env_rows if env_rows in ("auto", "all") else int(env_rows)
        self.decode_precision = os.environ.get("LMI_DECODE_PRECISION", "1") == "1"   # lo4 / split also cover the decode steps (decode_hl); 0 = fast decode (A/B)
        self._lo4_sel_cache: Dict[tuple, tuple] = {}
        self.skinny_fold_norm = True   # batched decode: RMSNorms folded into the projections (lmi_gemm_skinny_ex producer / consumer); False: norm launches
        self.skinny_packed = True      # batched decode over nn.Linear-layout weights (TP, pack_llm_weights=False): stream a packed second copy
        self.fp8_fused = True          # fp8 schedule: attention writes the fp8 o_proj operand, q|k|v GEMM does RoPE + KV append (False: separate launches)
        self.fp8_attention = os.environ.get("LMI_FP8_ATTENTION", "0") == "1"   # fp8 schedule: QK^T and PV of the Llama layers on the fp8 pipe too (attention_fp8.h)
        self._fp8 = None               # leopard_amd.fp8.Fp8Plan: fp8 operands for the ViT / LLM layer linears (enable_fp8; configs[4])
        self._rec = None               # calibration recorder callable((tower, layer, site), operand tensor)
        tc = cfg.text_config
        self._inv_freq = llama3_inv_freq(tc.head_dim, tc.rope_theta, tc.rope_scaling).to(self.device)
        self._geom_cache: Dict[tuple, tuple] = {}      # seq_lens -> (cu, cos, sin, last_rows) device tensors
        self._vit_cu_cache: Dict[int, torch.Tensor] = {}
        # ONE copy of the LLM weights (default; LMI_PACK_LLM_WEIGHTS=0 / pack_llm_weights=False keep the nn.Linear layout): see pack_llm_weights
        if pack_llm_weights is None:
            pack_llm_weights = os.environ.get("LMI_PACK_LLM_WEIGHTS", "1") == "1"
        if pack_llm_weights:
            self.pack_llm_weights()

    # ------------------------------------------------------------------------------------------------
    @property
    def fp8(self):
        return self._fp8

    @fp8.setter
    def fp8(self, plan):
        """Switching the schedule invalidates the captured vision-encode graphs (they replay the launches of the old one)."""
        self._fp8 = plan
        self._encode_graphs.clear()

    @property
    def tp_size(self) -> int:
        return getattr(self.W, "tp_size", 1)

    @property
    def precision(self) -> str:
        """"fast" (one rounding per operand hand-over: the benchmarked schedule up to round 4), "lo4" (+ the fp4 correction phase: meets
        north_star's 1e-3 at full depth), "split" (hi + lo 16-bit operand pairs at 2 K: the most exact, ~1.9 x the time)."""
        return "split" if self.split_operands else ("lo4" if self.lo4 else "fast")

    def lo4_supported(self) -> bool:
        """The lo4 schedule rides on the fused Llama / Mistral layer (head_dim 128, rope-ordered q|k|v rows, hidden % 256 == 0) and on
        32-element blocks along every contraction axis (hidden sizes and FFN widths % 32 == 0); 16-bit compute type.  Tensor-parallel
        engines run it too (round 5): the sequence-parallel norms hand over Lo4 pairs and the all-gathers move the images with the rows."""
        tc, vc, W = self.cfg.text_config, self.cfg.vision_config, self.W
        return bool(self.dtype in (torch.float16, torch.bfloat16) and tc.head_dim == 128 and tc.hidden_size % 256 == 0
                    and W.llm_layers and W.llm_layers[0].qkv_w_rope is not None and (tc.intermediate_size // self.tp_size) % 32 == 0
                    and tc.intermediate_size // self.tp_size >= 128 and vc.hidden_size % 32 == 0)

    @precision.setter
    def precision(self, mode: str):
        if mode not in ("fast", "lo4", "split"):
            raise ValueError(f"precision must be 'fast', 'lo4' or 'split', not {mode!r}")
        if mode == "split" and self.tp_size > 1:
            raise ValueError("the split-operand mode runs on one rank (tensor-parallel engines: 'fast' or 'lo4')")
        if mode == "lo4" and not self.lo4_supported():
            raise ValueError("precision 'lo4' needs the fused Llama / Mistral layer shape (head_dim 128, hidden % 256 == 0): use 'split'")
        self.split_operands, self.lo4 = mode == "split", mode == "lo4"
        self._encode_graphs.clear()                           # captured encodes replay the launches of the old schedule
        if self.lo4 and self.device.type == "cuda":
            self._lo4_weights("llm")                          # the multi-GB quantisation happens HERE, not inside the first (timed) prefill

    def decode_hl(self, B: int = 1) -> bool:
        """Decode precision mode (round 6): with ``precision`` = "lo4" / "split" the captured decode step hands every projection operand over as a
        pair of 16-bit rows — T(x) and T(x - T(x)) — through the M <= 16 kernels (lmi_gemm_skinny_hl ...): the hand-over roundings of the token's
        own path through the layers, which are what its logits' error is made of, are gone, at no extra weight traffic.  Needs the packed one-copy
        weight layout (the default on one rank), the folded norms and 2 B <= 16; otherwise the step is the fast one."""
        D = self.cfg.text_config.hidden_size
        return bool((self.lo4 or self.split_operands) and self.decode_precision and self.tp_size == 1 and self.llm_packed and 2 * B <= 16
                    and self.skinny_fold_norm and D % 16 == 0 and self.cfg.text_config.head_dim == 128
                    and all(L.qkv_w_rope is not None for L in self.W.llm_layers))

    def _llm_heads(self) -> Tuple[int, int]:
        """(query heads, kv heads) this rank computes."""
        tc = self.cfg.text_config
        return (getattr(self.W, "llm_heads", 0) or tc.num_attention_heads, getattr(self.W, "llm_kv_heads", 0) or tc.num_key_value_heads)

    # ---- one copy of the LLM weights ----------------------------------------------------------------------------------------------
    @property
    def llm_packed(self) -> bool:
        """True when the layer linears are stored in the packed order.  The layout of a tensor is a mark on the tensor object
        (weights.mark_packed), which a copy (.clone() / .to()) does not carry: the engine remembers what it packed and refuses to run on a
        weight set whose marks disagree with that, instead of reading a packed matrix as row-major."""
        from .weights import is_packed
        L0 = self.W.llm_layers[0] if self.W.llm_layers else None
        marked = L0 is not None and is_packed(L0.o_w)
        want = getattr(self.W, "_llm_packed", None)
        if want is not None and any(is_packed(getattr(L, n)) != want for L in self.W.llm_layers for n in ("qkv_w_rope", "o_w", "gu_w", "down_w")
                                    if getattr(L, n) is not None):                # every layer, every linear (128 attribute reads per pass)
            raise RuntimeError("LLM layer weights were replaced by copies that lost their layout mark (weights.mark_packed); "
                               "call engine.pack_llm_weights() / unpack_llm_weights() instead of copying packed tensors")
        return marked

    def pack_llm_weights(self) -> bool:
        """Store the Llama / Mistral layer linears ONCE, in the operand order the decode kernels stream (weights.skinny_pack), in place.
        The prefill GEMM stages its LDS image from that order too (ldw = LMI_LDW_PACKED(K): the packed order is a permutation of the
        16-byte pieces an LDS-DMA lane picks anyway — same image, same MFMA order, same bits, csrc/gemm.h GemmStager), the batch-1 decode
        step runs on lmi_gemm_skinny with one row, and the natural-order duplicate of q|k|v (1.6 GB for Llama-3.1-8B) is dropped:
        16.1 GB of layer weights are resident instead of 17.7 GB + a 15 GB second copy for batched decoding.  Single rank, head_dim 128,
        K % 128 == 0 (the conditions of the batched decode); returns False (and changes nothing) otherwise.  lm_head, embeddings and the
        vision side keep the nn.Linear layout."""
        from .weights import as_packed
        W = self.W
        if self.llm_packed:
            return True
        if not (self.tp_size == 1 and W.llm_layers and all(L.qkv_w_rope is not None for L in W.llm_layers) and self._batch_decode_supported()
                and self.dtype in (torch.float16, torch.bfloat16)):
            return False
        for L in W.llm_layers:
            for name in ("qkv_w_rope", "o_w", "gu_w", "down_w"):
                setattr(L, name, as_packed(getattr(L, name)))
            L.qkv_w = None
        W._llm_packed = True
        self._skinny_pack = None
        self._batch_states = {}                               # steps captured over a second copy of the weights
        return True

    def pack_vit_weights(self, packed: bool = True) -> int:
        """A/B knob (default: not applied): the SigLIP layer linears in the packed order as well — the GEMM's W staging reads 2-KiB-contiguous
        row groups instead of 128-byte row segments 2.3 / 8.7 KiB apart (the effect that made the Llama GEMMs 0.7 % faster).  Same bits.  Returns
        the number of tensors converted.  The fp8 plan and the split-operand copies are built from row-major views either way."""
        from .weights import as_packed, as_row_major, is_packed, packable
        n = 0
        for L in self.W.vit_layers:
            for name in ("qkv_w", "o_w", "fc1_w", "fc2_w"):
                w = getattr(L, name)
                if packed and not is_packed(w) and packable(w) and w.shape[0] % 128 == 0:
                    setattr(L, name, as_packed(w)); n += 1
                elif not packed and is_packed(w):
                    setattr(L, name, as_row_major(w)); n += 1
        if n:
            self._encode_graphs.clear()                       # captured encodes replay the launches of the old layout
        return n

    def unpack_llm_weights(self) -> None:
        """Back to the nn.Linear layout (A/B runs, tools that read the weights)."""
        from .weights import as_row_major
        if not self.llm_packed:
            return
        self.W._llm_packed = None                             # in transition
        for L in self.W.llm_layers:
            L.qkv_w = self._qkv_natural(L)
            for name in ("qkv_w_rope", "o_w", "gu_w", "down_w"):
                setattr(L, name, as_row_major(getattr(L, name)))
        self.W._llm_packed = False
        self._batch_states = {}                               # captured steps hold the packed tensors' launches (batch-1 states re-capture: _decode_run)
        self._head_pack = None

    def _qkv_natural(self, L) -> torch.Tensor:
        """q | k | v projection rows in their natural (checkpoint) order, row-major — kept beside the rope-ordered rows only while the
        weights are unpacked; rebuilt from them otherwise (rope_permute_rows is its own inverse)."""
        if L.qkv_w is not None:
            return L.qkv_w
        from .weights import as_row_major, rope_permute_rows
        (H, KV), hd = self._llm_heads(), self.cfg.text_config.head_dim
        w = as_row_major(L.qkv_w_rope)
        return torch.cat([rope_permute_rows(w[:(H + KV) * hd], hd), w[(H + KV) * hd:]], dim=0).contiguous()

    def _row_parallel(self, a: torch.Tensor, w: torch.Tensor, x: torch.Tensor, tmp: Optional[torch.Tensor]):
        """x += a @ w.T for o_proj / down_proj.  Single rank: fused in the GEMM's fp32 residual epilogue.  Tensor parallel: the
        rank's partial product goes to ``tmp`` (fp32), ONE all-reduce sums it over the ranks (RCCL over xGMI), then it is added."""
        if self.tp_size == 1:
            self.ops.gemm(a, w, x, epilogue=_lib.EPI_RESIDUAL)
            return
        self.ops.gemm(a, w, tmp, epilogue=_lib.EPI_STORE_F32)
        self.comm.all_reduce(tmp)
        x.add_(tmp)

    def _empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.device)

    def _carve(self, which: str, total: int, offsets, specs):
        """Views into the engine's caller-owned workspace of stage ``which`` ("llm" / "vit"; SURVEY.md 8b: sized by the library's
        lmi_*_workspace_bytes, owned by the caller): ONE uint8 allocation per stage AND LAUNCH STREAM, grown when a larger pass arrives and
        reused by every later one on that stream, carved at the byte offsets the library returned.  specs: [(rows, cols, dtype)] in LMI_WS_*
        order.  Per stream because passes on different HIP streams are not ordered against each other (bench.py --inflight > 1, a graph
        capture beside eager work): they must not share scratch.  A workspace is allocated while its stream is current, so the caching
        allocator's own stream bookkeeping covers its release."""
        if self._private_scratch:
            # warm-up and capture of a vision-encode graph (_encode_images_graph): inside capture the current stream is torch's capture stream,
            # whatever stream the graph is later launched on — a table keyed by it would hand every captured graph the SAME scratch, and
            # replays of two graphs on two launch streams (bench.py --inflight 2 --graph-encode) would race on it.  A graph therefore OWNS its
            # scratch: allocated here, inside the capture, from the graph's private pool; it lives and dies with the graph (round 6, advisor).
            ws = torch.empty(max(total, 256), dtype=torch.uint8, device=self.device)
        else:
            sid = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
            key = (which, sid)
            ws = self._workspaces.get(key)
            if ws is None or ws.numel() < total:
                if ws is None and len(self._workspaces) >= 16:          # streams come and go: bound the table.  An evicted workspace was allocated on
                    self._workspaces.pop(next(iter(self._workspaces)))  # ITS stream and only ever used there, so the allocator's reuse is ordered; no
                ws = self._workspaces[key] = torch.empty(max(total, 256), dtype=torch.uint8, device=self.device)   # graph points into the table
        out = []
        for off, (r, c, dt) in zip(offsets, specs):
            nbytes = r * c * torch.empty(0, dtype=dt).element_size()
            out.append(ws[off:off + nbytes].view(dt).view(r, c))
        return out

    def _pinned_to_device(self, t: torch.Tensor) -> torch.Tensor:
        """Host tensor -> device without blocking the host on the stream (pinned staging + async copy)."""
        if self.device.type != "cuda":
            return t.to(self.device)
        staged = t.pin_memory()
        out = staged.to(self.device, non_blocking=True)
        out._lmi_staging = staged                       # keep the pinned source alive until the copy has run
        return out

    def sequence_geometry(self, seq_lens: Sequence[int]):
        """cu_seqlens / RoPE tables / last-row indices for a tuple of packed sequence lengths (cached: the same
        prompt geometry recurs, and rebuilding would cost a blocking host->device copy per call)."""
        key = tuple(int(l) for l in seq_lens)
        hit = self._geom_cache.get(key)
        if hit is None:
            cu_list = [0]
            for l in key:
                cu_list.append(cu_list[-1] + l)
            cu = torch.tensor(cu_list, dtype=torch.int32, device=self.device)
            pos = torch.cat([torch.arange(l) for l in key])
            cos, sin = self.rope_tables(pos)
            last_rows = torch.tensor([c - 1 for c in cu_list[1:]], device=self.device)
            if len(self._geom_cache) > 64:
                self._geom_cache.clear()
            hit = self._geom_cache[key] = (cu, cos, sin, last_rows, cu_list)
        return hit

    def rope_tables(self, positions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos/sin [S, head_dim/2] fp32 on device (tiny; torch used as plumbing for a table build)."""
        f = positions.to(device=self.device, dtype=torch.float32).reshape(-1, 1) * self._inv_freq.reshape(1, -1)
        return f.cos().contiguous(), f.sin().contiguous()

    # ------------------------------------------------------------------------------------------------
    # a5 + a7: vision tower
    # ------------------------------------------------------------------------------------------------
    def lo4_vit_tiles(self, tiles_per_sample: Sequence[int], seq_lens: Sequence[int]) -> Optional[Tuple[bool, ...]]:
        """Per ViT input: does the lo4 schedule correct the SigLIP layer linears for it (``lo4_vit``: True / False / "auto" = the inputs of the
        samples whose LLM sequence has at most LO4_FULL_BELOW rows)?  None = no input is corrected (the fast tower)."""
        if not self.lo4 or self.lo4_vit is False:
            return None
        flags: List[bool] = []
        for n, S in zip(tiles_per_sample, seq_lens):
            flags += [bool(self.lo4_vit is True or int(S) <= self.LO4_FULL_BELOW)] * int(n)
        return tuple(flags) if any(flags) else None

    def vision_tower(self, tiles: torch.Tensor, lo4_tiles: Optional[Sequence[bool]] = None) -> torch.Tensor:
        """tiles: u8 [N,S,S,3] (HWC) or fp32 pixel_values [N,3,S,S].  Returns post-LN features T [N*T, D].  ``lo4_tiles``: per ViT input, whether
        the lo4 schedule corrects the tower's linears for it (lo4_vit_tiles; None = none)."""
        ops, W, vc = self.ops, self.W, self.cfg.vision_config
        n = tiles.shape[0]
        T, D, H, hd = vc.num_patches, vc.hidden_size, vc.num_attention_heads, vc.head_dim
        M = n * T
        x = self._empty(M, D, dtype=torch.float32)
        # normalise + im2col + patch conv + bias + position embedding: one launch, no im2col matrix in HBM (lmi_patch_embed)
        ops.patch_embed(tiles, W.patch_w_fused, W.patch_b, W.pos_emb, x, vc.image_size, vc.patch_size)
        cu = self._vit_cu_cache.get(n)
        if cu is None:
            cu = self._vit_cu_cache[n] = torch.arange(0, (n + 1) * T, T, dtype=torch.int32, device=self.device)
        if self.trace:
            self.trace("vit.embed", x)
        if self.fp8 is not None:
            return self._vit_layers_fp8(x, n)
        if self.split_operands:
            return self._vit_layers_split(x, n)
        if self.lo4 and lo4_tiles is not None and any(lo4_tiles):
            return self._vit_layers_lo4(x, n, lo4_tiles)
        qkv_w = W.vit_layers[0].qkv_w.shape[0] if W.vit_layers else 3 * D
        total, offs = ops.vit_workspace(M, D, qkv_w, W.vit_ff, self.dtype)
        h, qkv, att, ff = self._carve("vit", total, offs, [(M, D, self.dtype), (M, qkv_w, self.dtype), (M, D, self.dtype), (M, W.vit_ff, self.dtype)])
        scale = hd ** -0.5
        rec = self._rec
        for li, L in enumerate(W.vit_layers):
            ops.layernorm(x, L.ln1_w, L.ln1_b, h, vc.layer_norm_eps)
            rec and rec(("vit", li, "h1"), h)
            ops.gemm(h, L.qkv_w, qkv, bias=L.qkv_b)
            ops.attention(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], att, cu, cu, T, H, H, hd, scale, False,
                          self.use_tr)
            rec and rec(("vit", li, "att"), att)
            ops.gemm(att, L.o_w, x, bias=L.o_b, epilogue=_lib.EPI_RESIDUAL)
            ops.layernorm(x, L.ln2_w, L.ln2_b, h, vc.layer_norm_eps)
            rec and rec(("vit", li, "h2"), h)
            ops.gemm(h, L.fc1_w, ff, bias=L.fc1_b, act=_lib.ACT_GELU_TANH)
            rec and rec(("vit", li, "ff"), ff)
            ops.gemm(ff, L.fc2_w, x, bias=L.fc2_b, epilogue=_lib.EPI_RESIDUAL)
            if self.trace:
                self.trace(f"vit.{li}", x)
        out = self._empty(M, D)                 # a result, not scratch: the caller may hold it across later passes that reuse the workspace
        ops.layernorm(x, W.post_ln_w, W.post_ln_b, out, vc.layer_norm_eps)
        return out

    def _vit_layers_fp8(self, x: torch.Tensor, n: int) -> torch.Tensor:
        """The SigLIP layers with fp8 linears (leopard_amd.fp8): LayerNorm -> fp8 operand in one launch, fc1's GELU epilogue
        writes fc2's fp8 operand; q|k|v and the attention stay 16-bit, the residual stream fp32."""
        ops, W, vc, P = self.ops, self.W, self.cfg.vision_config, self.fp8
        T, D, H, hd = vc.num_patches, vc.hidden_size, vc.num_attention_heads, vc.head_dim
        M = n * T
        u8 = torch.uint8
        h8 = self._empty(M, D, dtype=u8)
        att8 = self._empty(M, D, dtype=u8)
        ff8 = self._empty(M, W.vit_ff, dtype=u8)
        qkv = self._empty(M, W.vit_layers[0].qkv_w.shape[0])
        att = self._empty(M, D)
        cu = self._vit_cu_cache[n]
        scale = hd ** -0.5
        fused = self.fp8_fused and self.use_tr          # (the fp8-output attention lives in the LDS-DMA kernel, the production one)
        for li, (L, Q) in enumerate(zip(W.vit_layers, P.vit)):
            ops.norm_fp8(x, L.ln1_w, L.ln1_b, h8, vc.layer_norm_eps, 2.0 ** Q.act["h1"])
            ops.gemm_fp8(h8, Q.lin["qkv"].w8, qkv, bias=L.qkv_b, scale_exp=Q.out_exp("h1", "qkv"))
            if fused:
                ops.attention_fp8out(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], att8, 2.0 ** Q.act["att"], cu, cu, T, H, H, hd, scale, False)
            else:
                ops.attention(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], att, cu, cu, T, H, H, hd, scale, False,
                              self.use_tr)
                ops.quantize_fp8(att, att8, 2.0 ** Q.act["att"])
            ops.gemm_fp8(att8, Q.lin["o"].w8, x, bias=L.o_b, epilogue=_lib.EPI_RESIDUAL, scale_exp=Q.out_exp("att", "o"))
            ops.norm_fp8(x, L.ln2_w, L.ln2_b, h8, vc.layer_norm_eps, 2.0 ** Q.act["h2"])
            ops.gemm_fp8(h8, Q.lin["fc1"].w8, ff8, bias=L.fc1_b, act=_lib.ACT_GELU_TANH, scale_exp=Q.out_exp("h2", "fc1"),
                         out_scale=2.0 ** Q.act["ff"])
            ops.gemm_fp8(ff8, Q.lin["fc2"].w8, x, bias=L.fc2_b, epilogue=_lib.EPI_RESIDUAL, scale_exp=Q.out_exp("ff", "fc2"))
            if self.trace:
                self.trace(f"vit.{li}", x)
        h = self._empty(M, D)
        ops.layernorm(x, W.post_ln_w, W.post_ln_b, h, vc.layer_norm_eps)
        return h

    # ---- split-operand precision mode -------------------------------------------------------------------------------------------
    def _split_weights(self):
        """[W | W] copies of the layer-linear weights (K doubled), built on first use: +0.8 GB (SigLIP) + 14 GB (Llama-3.1-8B)."""
        sw = getattr(self, "_split_w", None)
        if sw is None:
            from .weights import as_row_major
            dup = lambda w: (lambda r: torch.cat([r, r], dim=1).contiguous())(as_row_major(w))
            W = self.W
            sw = self._split_w = {
                "vit": [(dup(L.qkv_w), dup(L.o_w), dup(L.fc1_w), dup(L.fc2_w)) for L in W.vit_layers],
                "llm": [(dup(L.qkv_w_rope if L.qkv_w_rope is not None else L.qkv_w), dup(L.o_w), dup(L.gu_w), dup(L.down_w)) for L in W.llm_layers]}
        return sw

    def _vit_layers_split(self, x: torch.Tensor, n: int) -> torch.Tensor:
        """The SigLIP layers with split (hi + lo) A operands: LayerNorm -> fp32 -> [hi | lo]; attention output and GELU output in fp32 ->
        [hi | lo]; every linear at 2 K against [W | W].  q / k / v and the attention arithmetic stay 16-bit."""
        ops, W, vc = self.ops, self.W, self.cfg.vision_config
        T, D, H, hd = vc.num_patches, vc.hidden_size, vc.num_attention_heads, vc.head_dim
        M = n * T
        f32 = torch.float32
        h32, h2 = self._empty(M, D, dtype=f32), self._empty(M, 2 * D)
        qkv = self._empty(M, W.vit_layers[0].qkv_w.shape[0])
        ff32, ff2 = self._empty(M, W.vit_ff, dtype=f32), self._empty(M, 2 * W.vit_ff)
        cu = self._vit_cu_cache[n]
        scale = hd ** -0.5
        for li, (L, (qkv_w2, o_w2, fc1_w2, fc2_w2)) in enumerate(zip(W.vit_layers, self._split_weights()["vit"])):
            ops.layernorm(x, L.ln1_w, L.ln1_b, h32, vc.layer_norm_eps)
            ops.split_hi_lo(h32, h2)
            ops.gemm(h2, qkv_w2, qkv, bias=L.qkv_b)
            ops.attention_f32out(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], h32, cu, cu, T, H, H, hd, scale, False)
            ops.split_hi_lo(h32, h2)
            ops.gemm(h2, o_w2, x, bias=L.o_b, epilogue=_lib.EPI_RESIDUAL)
            ops.layernorm(x, L.ln2_w, L.ln2_b, h32, vc.layer_norm_eps)
            ops.split_hi_lo(h32, h2)
            ops.gemm(h2, fc1_w2, ff32, bias=L.fc1_b, act=_lib.ACT_GELU_TANH, epilogue=_lib.EPI_STORE_F32)
            ops.split_hi_lo(ff32, ff2)
            ops.gemm(ff2, fc2_w2, x, bias=L.fc2_b, epilogue=_lib.EPI_RESIDUAL)
            if self.trace:
                self.trace(f"vit.{li}", x)
        h = self._empty(M, D)
        ops.layernorm(x, W.post_ln_w, W.post_ln_b, h, vc.layer_norm_eps)
        return h

    def _llm_layers_split(self, x, cache, cu, cos, sin, max_len):
        """The Llama / Mistral layers with split (hi + lo) A operands (see _vit_layers_split); q|k|v + RoPE + KV append stay one launch
        (lmi_rmsnorm_rope on the 2 K operand), the attention hands over fp32."""
        ops, W, tc = self.ops, self.W, self.cfg.text_config
        S, D = x.shape
        (H, KV), hd = self._llm_heads(), tc.head_dim
        qw, kw = H * hd, KV * hd
        f32 = torch.float32
        h32, h2 = self._empty(S, D, dtype=f32), self._empty(S, 2 * D)
        a32, a2 = (h32, h2) if qw == D else (self._empty(S, qw, dtype=f32), self._empty(S, 2 * qw))
        qkv = self._empty(S, qw + 2 * kw)
        gu32, gu2 = self._empty(S, W.llm_ff, dtype=f32), self._empty(S, 2 * W.llm_ff)
        scale = hd ** -0.5
        for i, (L, (qkv_w2, o_w2, gu_w2, down_w2)) in enumerate(zip(W.llm_layers, self._split_weights()["llm"])):
            ops.rmsnorm(x, L.in_norm, h32, tc.rms_norm_eps)
            ops.split_hi_lo(h32, h2)
            if L.qkv_w_rope is not None and hd == 128:
                ops.rmsnorm_rope(h2, qkv_w2, qkv, None, tc.rms_norm_eps, cos, sin, cache.k[i] if cache else None, cache.v[i] if cache else None,
                                 0, H, KV, hd)
            else:
                ops.gemm(h2, qkv_w2, qkv)
                ops.rope_qk(qkv, H, KV, hd, cos, sin, cache.k[i] if cache else None, cache.v[i] if cache else None, 0)
            ops.attention_f32out(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], a32, cu, cu, max_len, H, KV, hd, scale, True,
                                 window=tc.sliding_window or 0)
            ops.split_hi_lo(a32, a2)
            ops.gemm(a2, o_w2, x, epilogue=_lib.EPI_RESIDUAL)
            ops.rmsnorm(x, L.post_norm, h32, tc.rms_norm_eps)
            ops.split_hi_lo(h32, h2)
            ops.gemm(h2, gu_w2, gu32, epilogue=_lib.EPI_SWIGLU_F32)
            ops.split_hi_lo(gu32, gu2)
            ops.gemm(gu2, down_w2, x, epilogue=_lib.EPI_RESIDUAL)
            if self.trace:
                self.trace(f"llm.{i}", x)

    # ---- low-bit correction mode --------------------------------------------------------------------------------------------------------
    LO4_FULL_BELOW = 1024              # "auto": sequences up to this length carry the correction on every row ...
    LO4_TAIL_ROWS = 16                 # ... longer ones on their last LO4_TAIL_ROWS rows: the row whose logits are read + a margin of 15 (tools/lo4_policy_study.py,
                                       # profiles/r06_lo4_policy_study_*.txt; on the device the last 1 / 16 / 64 / 256 / 1024 rows and every row all land at 2.4 - 2.8e-4 on
                                       # C3 and 3.7 - 4.1e-4 on C2).  16 rows sit in ONE 256-row tile 15 times out of 16, so one row tile per sequence runs the fp4 k-tiles;
                                       # 256 rows always straddled two (C2: 32.7 -> 31.6 ms, C3: 137.4 -> 137.0 ms on one box, profiles/r06_lo4_tail_rows_ab.txt)

    def _lo4_weights(self, tower: str = "llm"):
        """fp4 images (+ one E8M0 scale per row) of one tower's layer-linear weights, built from the row-major order of each weight when the
        mode is selected (or on first use): + 0.5 B per parameter (3.5 GB Llama-3.1-8B; 0.2 GB SigLIP, only with ``lo4_vit``)."""
        lw = self._lo4_w.get(tower)
        if lw is None:
            from .weights import as_row_major
            q = lambda w, head_pad=None: self.ops.quantize_w4(as_row_major(w).contiguous(), head_pad=head_pad)
            W, vc = self.W, self.cfg.vision_config
            if tower == "vit":
                vhp = (vc.num_attention_heads, vc.head_dim)      # out_proj's image in the per-head padded k order of the attention's residual image
                lw = [(q(L.qkv_w), q(L.o_w, vhp), q(L.fc1_w), q(L.fc2_w)) for L in W.vit_layers]
            else:
                lw = [(q(L.qkv_w_rope if L.qkv_w_rope is not None else L.qkv_w), q(L.o_w), q(L.gu_w), q(L.down_w)) for L in W.llm_layers]
            self._lo4_w[tower] = lw
        return lw

    def invalidate_lo4_weights(self) -> None:
        """Drop the fp4 weight images: call after replacing or changing the VALUES of layer weights (the images are a function of the values; a
        change of layout — pack / unpack — keeps them valid)."""
        self._lo4_w = {}

    def lo4_tail_rows(self, seq_len: int) -> int:
        """How many trailing rows of a ``seq_len``-row sequence carry the correction under ``lo4_rows``."""
        r = self.lo4_rows
        if r == "all":
            return seq_len
        if r == "auto":
            return seq_len if seq_len <= self.LO4_FULL_BELOW else min(self.LO4_TAIL_ROWS, seq_len)
        return max(1, min(int(r), seq_len))

    def _lo4_selection(self, seq_lens: Sequence[int]):
        """(row_sel uint8 [S], unit_sel uint8 [ceil(S / 64)]) device tensors + the host [n, 2] int32 array of the selected row ranges (the
        tile-order hint of lmi_lo4.sel_ranges) for the packed rows of ``seq_lens``, or None when every row is selected."""
        tails = [self.lo4_tail_rows(int(l)) for l in seq_lens]
        if all(t == int(l) for t, l in zip(tails, seq_lens)):
            return None
        key = (tuple(int(l) for l in seq_lens), tuple(tails))
        hit = self._lo4_sel_cache.get(key)
        if hit is None:
            S = int(sum(key[0]))
            row = np.zeros(S, dtype=np.uint8)
            end, ranges = 0, []
            for l, t in zip(*key):
                end += l
                row[end - t:end] = 1
                if ranges and ranges[-1][1] == end - t:
                    ranges[-1][1] = end
                else:
                    ranges.append([end - t, end])
            unit = np.zeros((S + 63) // 64 * 64, dtype=np.uint8)
            unit[:S] = row
            unit = unit.reshape(-1, 64).max(axis=1)
            if len(self._lo4_sel_cache) >= 64:
                self._lo4_sel_cache.pop(next(iter(self._lo4_sel_cache)))
            hit = self._lo4_sel_cache[key] = (self._pinned_to_device(torch.from_numpy(row)), self._pinned_to_device(torch.from_numpy(unit)),
                                              np.ascontiguousarray(np.array(ranges, dtype=np.int32).reshape(-1, 2)))
        return hit

    def _lo4_act(self, rows: int, width: int, heads: Optional[tuple] = None, sel: Optional[tuple] = None):
        """Operand pair buffers; ``heads`` = (n_heads, head_dim): an attention output (image in the per-head padded k order); ``sel``: the row
        selection of the pass (_lo4_selection) — the images then start out zero and only selected rows are ever written."""
        from .ops import Lo4Act, lo4_head_k4
        return Lo4Act.empty(rows, width, self.dtype, self.device, k4=lo4_head_k4(*heads) if heads else None, sel=sel)

    def _vit_layers_lo4(self, x: torch.Tensor, n: int, flags: Optional[Sequence[bool]] = None) -> torch.Tensor:
        """The SigLIP layers with the low-bit correction phase: the LayerNorms and fc1's GELU epilogue hand over T(y) + the fp4 image of
        y - T(y) directly, and so does the attention kernel (lmi_attn_varlen_fwd_lo4: every head padded to 96 slots in the image, out_proj's
        weight image laid out to match).  q / k / v and the attention arithmetic stay 16-bit."""
        ops, W, vc = self.ops, self.W, self.cfg.vision_config
        T, D, H, hd = vc.num_patches, vc.hidden_size, vc.num_attention_heads, vc.head_dim
        M = n * T
        # (a packed batch whose samples differ: the row selection of the correction phase carries the per-input decision — every row of a selected
        # ViT input, none of the others: _lo4_selection with "sequences" = the inputs)
        sel = None
        if flags is not None and not all(flags):
            key = ("vit", tuple(bool(f) for f in flags))
            sel = self._lo4_sel_cache.get(key)
            if sel is None:
                row = np.repeat(np.array(flags, dtype=np.uint8), T)
                unit = np.zeros((M + 63) // 64 * 64, dtype=np.uint8)
                unit[:M] = row
                r = np.flatnonzero(np.diff(np.concatenate([[0], row.astype(np.int8), [0]])))
                sel = self._lo4_sel_cache[key] = (self._pinned_to_device(torch.from_numpy(row)), self._pinned_to_device(torch.from_numpy(unit.reshape(-1, 64).max(axis=1))),
                                                  np.ascontiguousarray(r.reshape(-1, 2).astype(np.int32)))
        h, att, ff = self._lo4_act(M, D, sel=sel), self._lo4_act(M, D, heads=(H, hd), sel=sel), self._lo4_act(M, W.vit_ff, sel=sel)
        qkv = self._empty(M, W.vit_layers[0].qkv_w.shape[0])
        cu = self._vit_cu_cache[n]
        scale = hd ** -0.5
        for li, (L, (qkv4, o4, fc14, fc24)) in enumerate(zip(W.vit_layers, self._lo4_weights("vit"))):
            ops.norm_lo4(x, L.ln1_w, L.ln1_b, h, vc.layer_norm_eps)
            ops.gemm_lo4(h, L.qkv_w, qkv4, qkv, bias=L.qkv_b)
            ops.attention_lo4(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], att, cu, cu, T, H, H, hd, scale, False)
            ops.gemm_lo4(att, L.o_w, o4, x, bias=L.o_b, epilogue=_lib.EPI_RESIDUAL)
            ops.norm_lo4(x, L.ln2_w, L.ln2_b, h, vc.layer_norm_eps)
            ops.gemm_lo4(h, L.fc1_w, fc14, ff.hi, bias=L.fc1_b, act=_lib.ACT_GELU_TANH, out4=ff)
            ops.gemm_lo4(ff, L.fc2_w, fc24, x, bias=L.fc2_b, epilogue=_lib.EPI_RESIDUAL)
            if self.trace:
                self.trace(f"vit.{li}", x)
        out = self._empty(M, D)
        ops.layernorm(x, W.post_ln_w, W.post_ln_b, out, vc.layer_norm_eps)
        return out

    def _llm_layers_lo4(self, x, cache, cu, cos, sin, max_len, seq_lens=None, all_rows=False):
        """The Llama / Mistral layers with the low-bit correction phase, on the FUSED schedule of the fast path: the RMSNorms ride in the GEMM
        epilogues (the producers o_proj / down_proj also write the fp4 image of the residual of T(x gamma)), q|k|v + RoPE + KV append is one
        launch, gate/up's SwiGLU epilogue writes down_proj's operand pair, the attention kernel o_proj's: no launch is added to the fast schedule."""
        ops, W, tc = self.ops, self.W, self.cfg.text_config
        S, D = x.shape
        (H, KV), hd = self._llm_heads(), tc.head_dim
        qw, kw = H * hd, KV * hd
        if not (hd == 128 and D % 256 == 0 and W.llm_layers and W.llm_layers[0].qkv_w_rope is not None):
            raise RuntimeError("precision 'lo4' needs head_dim 128 and the rope-ordered q|k|v weights (the fused Llama / Mistral schedule)")
        # row selection (lo4_rows): only the trailing rows of each sequence hand over residual images; the other rows' images stay zero and the
        # tiles without a selected row skip the fp4 k-tiles (csrc/gemm.h GemmArgs::row_sel)
        # (``all_rows``: the caller reads the logits of EVERY row — all_logits — so every row is a logits row)
        sel = None if all_rows else self._lo4_selection(seq_lens if seq_lens is not None else [S])
        h, att, gu = self._lo4_act(S, D, sel=sel), self._lo4_act(S, qw, heads=(H, hd), sel=sel), self._lo4_act(S, W.llm_ff, sel=sel)
        qkv = self._empty(S, qw + 2 * kw)
        parts = (D + 63) // 64
        sq_a, sq_b = self._empty(S, parts, dtype=torch.float32), self._empty(S, parts, dtype=torch.float32)
        scale = hd ** -0.5
        n_layers = len(W.llm_layers)
        for i, (L, (qkv4, o4, gu4, down4)) in enumerate(zip(W.llm_layers, self._lo4_weights("llm"))):
            if i == 0:
                ops.norm_lo4(x, L.in_norm, None, h, tc.rms_norm_eps)
            ops.rmsnorm_rope_lo4(h, L.qkv_w_rope, qkv4, qkv, None if i == 0 else sq_b, tc.rms_norm_eps, cos, sin,
                                 cache.k[i] if cache else None, cache.v[i] if cache else None, 0, H, KV, hd)
            ops.attention_lo4(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], att, cu, cu, max_len, H, KV, hd, scale, True,
                              window=tc.sliding_window or 0)
            ops.gemm_lo4(att, L.o_w, o4, x, epilogue=_lib.EPI_RESIDUAL, norm_out=h.hi, norm_gamma=L.post_norm, rowsq_out=sq_a, out4=h)
            ops.gemm_lo4(h, L.gu_w, gu4, gu.hi, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq_a, norm_dim=D, norm_eps=tc.rms_norm_eps, out4=gu)
            if i + 1 < n_layers:
                ops.gemm_lo4(gu, L.down_w, down4, x, epilogue=_lib.EPI_RESIDUAL, norm_out=h.hi, norm_gamma=W.llm_layers[i + 1].in_norm,
                             rowsq_out=sq_b, out4=h)
            else:
                ops.gemm_lo4(gu, L.down_w, down4, x, epilogue=_lib.EPI_RESIDUAL)
            if self.trace:
                self.trace(f"llm.{i}", x)

    def enable_fp8(self, calibration_samples, headroom: float = 2.0):
        """Switch the ViT / LLM layer linears to fp8 operands (BASELINE configs[4]): quantise the weights once, take the static
        activation scales from a 16-bit prefill of ``calibration_samples`` [(input_ids, tiles)].  ``engine.fp8 = None`` reverts."""
        from . import fp8 as F8
        self.fp8 = F8.calibrate(self, calibration_samples, headroom)
        return self.fp8

    # ------------------------------------------------------------------------------------------------
    # a8 + a9: pixel shuffle + projector
    # ------------------------------------------------------------------------------------------------
    def project(self, vit_out: torch.Tensor, n_tiles: int) -> torch.Tensor:
        """T [N*T, Dv] -> visual tokens fp32 [N*T/4, Dt] (pixel shuffle folded into linear_1's A gather)."""
        ops, W, cfg = self.ops, self.W, self.cfg
        rows = n_tiles * cfg.tokens_per_tile
        Dt = cfg.text_config.hidden_size
        h1 = self._empty(rows, Dt)
        ops.gemm(vit_out, W.proj1_w, h1, bias=W.proj1_b, act=_lib.ACT_GELU_ERF, a_mode=_lib.A_PIXEL_SHUFFLE,
                 ps_grid=cfg.vision_config.grid, M=rows)
        vis = self._empty(rows, Dt, dtype=torch.float32)
        ops.gemm(h1, W.proj2_w, vis, bias=W.proj2_b, epilogue=_lib.EPI_STORE_F32)
        return vis

    def encode_images(self, tiles: torch.Tensor, lo4_tiles: Optional[Sequence[bool]] = None) -> torch.Tensor:
        if self.graph_encode and self.device.type == "cuda" and not self.ops.emulated and tiles.dtype == torch.uint8:
            return self._encode_images_graph(tiles, lo4_tiles)
        return self.project(self.vision_tower(tiles, lo4_tiles), tiles.shape[0])

    def _encode_images_graph(self, tiles: torch.Tensor, lo4_tiles: Optional[Sequence[bool]] = None) -> torch.Tensor:
        """BASELINE config 5 ("hipGraph-captured encode"): the vision tower + projector for N ViT inputs is ~200 launches whose
        shapes depend on N only, so they are captured ONCE per N into a HIP graph over static buffers (u8 tiles in, fp32 visual
        tokens out) and replayed: one graph launch instead of ~200 stream launches, no per-launch host work.  Results are those
        of the eager path bit for bit (same kernels, same order).  The returned tensor is the graph's static output buffer:
        it is overwritten by the next encode of the same N (callers here consume it immediately in embed_merge)."""
        n = tiles.shape[0]
        # one graph + static buffers per (N, launch stream): replays of one graph from two streams (bench.py --inflight > 1) would
        # share the static buffers with nothing ordering them
        key = (n, torch.cuda.current_stream(self.device).cuda_stream, None if lo4_tiles is None else tuple(bool(f) for f in lo4_tiles))
        ent = self._encode_graphs.get(key)
        if ent is None:
            static_in = torch.empty_like(tiles)
            static_in.copy_(tiles)
            side = torch.cuda.Stream(device=self.device)                       # warm-up outside capture (LDS attributes, allocator)
            side.wait_stream(torch.cuda.current_stream(self.device))
            self._private_scratch = True                                       # the graph owns its tower scratch (see _carve)
            try:
                with torch.cuda.stream(side):
                    self.project(self.vision_tower(static_in, lo4_tiles), n)
                torch.cuda.current_stream(self.device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    static_out = self.project(self.vision_tower(static_in, lo4_tiles), n)
            finally:
                self._private_scratch = False
            if len(self._encode_graphs) >= 8:                                  # a handful of distinct N per workload; bound the pools
                self._encode_graphs.pop(next(iter(self._encode_graphs)))
            ent = self._encode_graphs[key] = (g, static_in, static_out)
        g, static_in, static_out = ent
        static_in.copy_(tiles)
        g.replay()
        return static_out

    # ------------------------------------------------------------------------------------------------
    # a10: embedding gather + merge
    # ------------------------------------------------------------------------------------------------
    def embed_merge(self, input_ids: torch.Tensor, visual_tokens: Optional[torch.Tensor]) -> torch.Tensor:
        cfg = self.cfg
        # the index map is planned on the host from the token ids; ids that are already host-side (the tokenizer's
        # output, EVAL:443) cost no device sync at all
        ids_host_t = input_ids.detach().reshape(-1).to("cpu", torch.int64)
        n_rows = 0 if visual_tokens is None else visual_tokens.shape[0]
        src = plan_merge(ids_host_t.numpy(), cfg.image_token_index, n_rows, cfg.tokens_per_tile)
        if input_ids.device == self.device:
            ids_dev = input_ids.reshape(-1).to(torch.int64).contiguous()
        else:
            ids_dev = self._pinned_to_device(ids_host_t.contiguous())
        src_dev = self._pinned_to_device(torch.from_numpy(src))
        x = self._empty(len(src), cfg.text_config.hidden_size, dtype=torch.float32)
        self.ops.embed_merge(ids_dev, src_dev, self.W.embed, visual_tokens, x)
        return x

    # ------------------------------------------------------------------------------------------------
    # a11: LLM prefill over one or several packed causal sequences
    # ------------------------------------------------------------------------------------------------
    def llm_prefill(self, x: torch.Tensor, seq_lens: Sequence[int], cache: Optional[KVCache] = None,
                    all_logits: bool = False):
        """x: fp32 [sum(seq_lens), D] residual stream (updated in place).  Returns (logits_last [n_seq, V],
        logits_all or None).  ``cache`` (single sequence only) receives rotated K and V."""
        ops, W, tc = self.ops, self.W, self.cfg.text_config
        S, D = x.shape
        (H, KV), hd = self._llm_heads(), tc.head_dim
        qw, kw = H * hd, KV * hd
        cu, cos, sin, last_rows, cu_list = self.sequence_geometry(seq_lens)
        assert cu_list[-1] == S
        self.llm_packed                                   # raises when a packed weight was replaced by a copy without its layout mark
        if cache is not None:       # one sequence, or a pool holding the packed rows of several (generate_batch splits it afterwards)
            assert cache.length == 0 and cache.capacity >= S
        max_len = max(int(l) for l in seq_lens)
        if self.trace:
            self.trace("llm.embed", x)
        if self.fp8 is not None or ((self.split_operands or self.lo4) and self.tp_size == 1):
            if self.fp8 is not None:
                self._llm_layers_fp8(x, cache, cu, cos, sin, max_len, seq_lens)
            elif self.lo4:
                self._llm_layers_lo4(x, cache, cu, cos, sin, max_len, seq_lens, all_rows=all_logits)
            else:
                self._llm_layers_split(x, cache, cu, cos, sin, max_len)
            if cache is not None:
                cache.length = S
            return self._lm_head(x, last_rows, all_logits)
        parts = (D + 63) // 64
        total, offs = ops.llm_prefill_workspace(S, D, H, KV, hd, W.llm_ff, self.dtype)
        h, qkv, att, gu, sq_a, sq_b = self._carve("llm", total, offs, [(S, D, self.dtype), (S, qw + 2 * kw, self.dtype), (S, qw, self.dtype),
                                                                         (S, W.llm_ff, self.dtype), (S, parts, torch.float32), (S, parts, torch.float32)])
        tmp = self._empty(S, D, dtype=torch.float32) if self.tp_size > 1 else None
        scale = hd ** -0.5
        # Fused schedule (one rank, head_dim 128): the RMSNorms and the RoPE ride in the GEMM epilogues.  Each residual GEMM
        # (o_proj, down_proj) also emits T(x * gamma_next) and per-row partial sums of squares; the GEMM that consumes them
        # (gate/up, next layer's qkv) applies rstd to its accumulator rows; the qkv GEMM rotates q / k and appends K / V to the
        # cache in its epilogue.  Only the very first norm of the stack is a launch of its own.
        fused = (self.fuse_norm_rope and self.tp_size == 1 and hd == 128 and D % 256 == 0 and W.llm_layers
                 and W.llm_layers[0].qkv_w_rope is not None)
        rec = self._rec
        if fused:
            n_layers = len(W.llm_layers)                            # sq_a: partials feeding gate/up; sq_b: feeding the next layer's qkv
            for i, L in enumerate(W.llm_layers):
                if i == 0:
                    ops.rmsnorm(x, L.in_norm, h, tc.rms_norm_eps)
                ops.rmsnorm_rope(h, L.qkv_w_rope, qkv, None if i == 0 else sq_b, tc.rms_norm_eps, cos, sin,
                                 cache.k[i] if cache else None, cache.v[i] if cache else None, 0, H, KV, hd)
                ops.attention(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], att, cu, cu, max_len, H, KV, hd, scale,
                              True, self.use_tr, window=tc.sliding_window or 0)
                ops.gemm_ex(att, L.o_w, x, epilogue=_lib.EPI_RESIDUAL, norm_out=h, norm_gamma=L.post_norm, rowsq_out=sq_a)
                ops.gemm_ex(h, L.gu_w, gu, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq_a, norm_dim=D, norm_eps=tc.rms_norm_eps)
                if i + 1 < n_layers:
                    ops.gemm_ex(gu, L.down_w, x, epilogue=_lib.EPI_RESIDUAL, norm_out=h, norm_gamma=W.llm_layers[i + 1].in_norm,
                                rowsq_out=sq_b)
                else:
                    ops.gemm(gu, L.down_w, x, epilogue=_lib.EPI_RESIDUAL)
                if self.trace:
                    self.trace(f"llm.{i}", x)
        else:
            for i, L in enumerate(W.llm_layers):
                ops.rmsnorm(x, L.in_norm, h, tc.rms_norm_eps)
                rec and rec(("llm", i, "h1"), h)
                if L.qkv_w is not None:
                    ops.gemm(h, L.qkv_w, qkv)
                    ops.rope_qk(qkv, H, KV, hd, cos, sin, cache.k[i] if cache else None, cache.v[i] if cache else None, 0)
                else:                                    # packed weights keep the rope-ordered rows only: projection + RoPE + KV append on the normalised rows
                    ops.rmsnorm_rope(h, L.qkv_w_rope, qkv, None, tc.rms_norm_eps, cos, sin, cache.k[i] if cache else None,
                                     cache.v[i] if cache else None, 0, H, KV, hd)
                if rec:                                  # operands of the fp8 attention arithmetic (rotated q / k, v)
                    rec(("llm", i, "q"), qkv[:, :qw]); rec(("llm", i, "k"), qkv[:, qw:qw + kw]); rec(("llm", i, "v"), qkv[:, qw + kw:])
                ops.attention(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], att, cu, cu, max_len, H, KV, hd, scale,
                              True, self.use_tr, window=tc.sliding_window or 0)
                rec and rec(("llm", i, "att"), att)
                self._row_parallel(att, L.o_w, x, tmp)
                ops.rmsnorm(x, L.post_norm, h, tc.rms_norm_eps)
                rec and rec(("llm", i, "h2"), h)
                ops.gemm(h, L.gu_w, gu, epilogue=_lib.EPI_SWIGLU)
                rec and rec(("llm", i, "gu"), gu)
                self._row_parallel(gu, L.down_w, x, tmp)
                if self.trace:
                    self.trace(f"llm.{i}", x)
        if cache is not None:
            cache.length = S
        return self._lm_head(x, last_rows, all_logits)

    def _llm_layers_fp8(self, x, cache, cu, cos, sin, max_len, seq_lens=None):
        """The Llama layers with fp8 linears (leopard_amd.fp8): RMSNorm -> fp8 operand in one launch, the SwiGLU epilogue of
        gate/up writes down_proj's fp8 operand; q|k|v, RoPE and the KV cache stay 16-bit, the stream fp32.  ``fp8_attention``: the
        attention's two products run on the fp8 matrix pipe as well (lmi_attn_prep_fp8 + lmi_attn_fp8_fwd: e4m3 q, k, v and P with the
        static scales of the plan; the cache keeps the 16-bit K / V for the decode)."""
        ops, W, tc, P = self.ops, self.W, self.cfg.text_config, self.fp8
        S, D = x.shape
        (H, KV), hd = self._llm_heads(), tc.head_dim
        qw, kw = H * hd, KV * hd
        u8 = torch.uint8
        h8 = self._empty(S, D, dtype=u8)
        att8 = self._empty(S, qw, dtype=u8)
        gu8 = self._empty(S, W.llm_ff, dtype=u8)
        qkv = self._empty(S, qw + 2 * kw)
        att = self._empty(S, qw)
        scale = hd ** -0.5
        fused = self.fp8_fused and self.use_tr          # (the fp8-output attention lives in the LDS-DMA kernel, the production one)
        a8 = (self.fp8_attention and fused and hd == 128 and seq_lens is not None and not (tc.sliding_window or 0)
              and all(k in Q.act for Q in P.llm for k in ("q", "k", "v")))
        if a8:
            tiles = [(int(l) + 63) // 64 for l in seq_lens]
            key = ("a8", tuple(int(l) for l in seq_lens))
            tb = self._geom_cache.get(key)
            if tb is None:
                tb = self._geom_cache[key] = self._pinned_to_device(torch.tensor([0] + list(np.cumsum(tiles)), dtype=torch.int32))
            n_tiles = sum(tiles)
            q8 = self._empty(S, qw, dtype=u8)
            k_img, v_img = self._empty(KV * n_tiles * 8192, dtype=u8), self._empty(KV * n_tiles * 8192, dtype=u8)
        for i, (L, Q) in enumerate(zip(W.llm_layers, P.llm)):
            ops.norm_fp8(x, L.in_norm, None, h8, tc.rms_norm_eps, 2.0 ** Q.act["h1"])
            if fused and hd == 128 and "qkv_rope" in Q.lin:
                ops.rope_qkv_fp8(h8, Q.lin["qkv_rope"].w8, qkv, Q.out_exp("h1", "qkv"), cos, sin, cache.k[i] if cache else None,
                                 cache.v[i] if cache else None, 0, H, KV, hd)
            else:
                ops.gemm_fp8(h8, Q.lin["qkv"].w8, qkv, scale_exp=Q.out_exp("h1", "qkv"))
                ops.rope_qk(qkv, H, KV, hd, cos, sin, cache.k[i] if cache else None, cache.v[i] if cache else None, 0)
            if a8:
                sq, sk, sv = 2.0 ** Q.act["q"], 2.0 ** Q.act["k"], 2.0 ** Q.act["v"]
                ops.attn_prep_fp8(qkv, cu, tb, n_tiles, H, KV, hd, sq, sk, sv, q8, k_img, v_img)
                ops.attention_fp8(q8, k_img, v_img, att8, cu, tb, n_tiles, max_len, H, KV, hd, scale, sq, sk, sv, causal=True,
                                  out_fp8_scale=2.0 ** Q.act["att"], dtype=self.dtype)
            elif fused:
                ops.attention_fp8out(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], att8, 2.0 ** Q.act["att"], cu, cu, max_len, H, KV, hd, scale,
                                     True, window=tc.sliding_window or 0)
            else:
                ops.attention(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], att, cu, cu, max_len, H, KV, hd, scale,
                              True, self.use_tr, window=tc.sliding_window or 0)
                ops.quantize_fp8(att, att8, 2.0 ** Q.act["att"])
            ops.gemm_fp8(att8, Q.lin["o"].w8, x, epilogue=_lib.EPI_RESIDUAL, scale_exp=Q.out_exp("att", "o"))
            ops.norm_fp8(x, L.post_norm, None, h8, tc.rms_norm_eps, 2.0 ** Q.act["h2"])
            ops.gemm_fp8(h8, Q.lin["gu"].w8, gu8, epilogue=_lib.EPI_SWIGLU, scale_exp=Q.out_exp("h2", "gu"),
                         out_scale=2.0 ** Q.act["gu"])
            ops.gemm_fp8(gu8, Q.lin["down"].w8, x, epilogue=_lib.EPI_RESIDUAL, scale_exp=Q.out_exp("gu", "down"))
            if self.trace:
                self.trace(f"llm.{i}", x)

    def _lm_head(self, x, last_rows, all_logits):
        ops, W, tc = self.ops, self.W, self.cfg.text_config
        V = tc.vocab_size
        # last position of every packed sequence: final RMSNorm + head in ONE launch over the fp32 rows (no 16-bit
        # rounding of the normalised row: lmi_lm_head_last)
        logits_last = self._empty(last_rows.numel(), W.lm_head.shape[0], dtype=torch.float32)
        ops.lm_head_last(W.lm_head, x, last_rows, W.final_norm, tc.rms_norm_eps, logits_last)
        logits_all = None
        if all_logits:
            hall = self._empty(*x.shape)
            ops.rmsnorm(x, W.final_norm, hall, tc.rms_norm_eps)
            logits_all = self._empty(x.shape[0], W.lm_head.shape[0], dtype=torch.float32)
            ops.gemm(hall, W.lm_head, logits_all, epilogue=_lib.EPI_STORE_F32)
            logits_all = logits_all[:, :V]
        return logits_last[:, :V], logits_all

    # ------------------------------------------------------------------------------------------------
    # the whole prefill for one sample (EVAL:261-333)
    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def prefill(self, input_ids: torch.Tensor, tiles: Optional[torch.Tensor], cache: Optional[KVCache] = None,
                all_logits: bool = False, keep_parts: bool = False, visual_tokens: Optional[torch.Tensor] = None
                ) -> PrefillResult:
        if self.tp_size > 1:
            if all_logits or keep_parts:
                raise NotImplementedError("all_logits / keep_parts are single-rank diagnostics")
            return self._prefill_tp(input_ids, tiles, cache, visual_tokens)
        parts = {} if keep_parts else None
        n_tiles = 0
        if visual_tokens is None and tiles is not None and tiles.shape[0] > 0:
            n_tiles = tiles.shape[0]
            vflags = self.lo4_vit_tiles([n_tiles], [input_ids.numel() + n_tiles * (self.cfg.tokens_per_tile - 1)])
            if keep_parts:
                vit = self.vision_tower(tiles, vflags)
                visual_tokens = self.project(vit, n_tiles)
                parts["vit"] = vit
            else:
                visual_tokens = self.encode_images(tiles, vflags)
        elif visual_tokens is not None:
            n_tiles = visual_tokens.shape[0] // self.cfg.tokens_per_tile
        if keep_parts and visual_tokens is not None:
            parts["visual_tokens"] = visual_tokens
        x = self.embed_merge(input_ids, visual_tokens)
        if keep_parts:
            parts["inputs_embeds"] = x.clone()
        S = x.shape[0]
        last, all_ = self.llm_prefill(x, [S], cache=cache, all_logits=all_logits)
        return PrefillResult(logits_last=last[0], seq_len=S, n_tiles=n_tiles, logits_all=all_, parts=parts)


    # ------------------------------------------------------------------------------------------------
    # ONE sample on all ranks (SURVEY.md 8e phases A + B): tile-sharded vision encode + one all-gather, then the
    # LLM tensor-parallel with sequence-parallel norms.  The fp32 residual stream is sharded by rows: with R ranks and NC row
    # chunks, chunk c holds global rows [c*R*Sl, (c+1)*R*Sl) and rank r owns rows [c*R*Sl + r*Sl, +Sl) of it.  Per half layer
    # and chunk:  RMSNorm(own rows) -> all-gather (16-bit, [R*Sl, D]) -> column-parallel GEMM(s) on all rows of the chunk for
    # this rank's heads / FFN slice -> row-parallel GEMM -> reduce-scatter (16-bit) -> residual add on the own rows.
    # Collectives go to a side stream; chunk c's exchange runs under chunk c+1's GEMMs.  Causal attention makes the row chunks
    # independent in the right order: chunk c attends to the K/V rows of chunks <= c, which the cache already holds.
    # ------------------------------------------------------------------------------------------------
    def _tp_geometry(self, S: int):
        R, NC = self.tp_size, max(1, int(self.tp_chunks))
        Sl = -(-S // (R * NC))
        Sl = (Sl + 7) // 8 * 8                                  # 16-byte aligned row blocks for the collectives
        return R, NC, Sl, R * Sl

    def tp_padded_len(self, S: int) -> int:
        """KV-cache rows a tensor-parallel prefill of S tokens writes (the row chunks are padded to equal size)."""
        R, NC, Sl, Sc = self._tp_geometry(S)
        return NC * Sc

    @torch.no_grad()
    def _prefill_tp(self, input_ids, tiles, cache, visual_tokens):
        from . import dist as D
        cfg, ops, W, tc = self.cfg, self.ops, self.W, self.cfg.text_config
        comm, dev = self.comm, self.device
        if visual_tokens is None and tiles is not None and tiles.shape[0] > 0:
            visual_tokens = D.encode_images_sharded(self, tiles)
        n_tiles = 0 if visual_tokens is None else visual_tokens.shape[0] // cfg.tokens_per_tile
        ids_host = input_ids.detach().reshape(-1).to("cpu", torch.int64)
        n_rows = 0 if visual_tokens is None else visual_tokens.shape[0]
        src = plan_merge(ids_host.numpy(), cfg.image_token_index, n_rows, cfg.tokens_per_tile)
        S = len(src)
        R, NC, Sl, Sc = self._tp_geometry(S)
        rank = comm.rank
        if cache is None:
            cache = KVCache(cfg, NC * Sc, self.dtype, dev, tp_size=R)
        assert cache.length == 0 and cache.capacity >= NC * Sc, "tensor-parallel prefill needs cache capacity >= tp_padded_len(S)"
        # merged embeddings of the rows this rank owns, chunk by chunk (rows past S: any text row, finite and never used)
        ids_dev = self._pinned_to_device(ids_host.contiguous()) if input_ids.device != dev else input_ids.reshape(-1).to(torch.int64).contiguous()
        pad_src = int(np.nonzero(src >= 0)[0][0]) if (src >= 0).any() else 0
        D_ = tc.hidden_size
        xs = []
        for c in range(NC):
            g0 = c * Sc + rank * Sl
            loc = np.full(Sl, src[pad_src], dtype=np.int64)
            n_real = max(0, min(Sl, S - g0))
            loc[:n_real] = src[g0:g0 + n_real]
            x = self._empty(Sl, D_, dtype=torch.float32)
            ops.embed_merge(ids_dev, self._pinned_to_device(torch.from_numpy(loc)), W.embed, visual_tokens, x)
            xs.append(x)
        self._llm_layers_tp(xs, S, cache)
        cache.length = S
        # last real row -> every rank (fp32, 16 KB), then the column-parallel head: rank r computes its slice of the vocabulary
        last = S - 1
        c_l, r_l, l_l = last // Sc, (last % Sc) // Sl, last % Sl
        xrow = xs[c_l][l_l:l_l + 1].clone() if rank == r_l else self._empty(1, D_, dtype=torch.float32)
        comm.broadcast(xrow, r_l)
        Vp = W.lm_head.shape[0]
        Vl = -(-Vp // R)
        lo, hi = min(Vp, rank * Vl), min(Vp, (rank + 1) * Vl)
        mine = torch.zeros(1, Vl, dtype=torch.float32, device=dev)
        if hi > lo:
            ops.lm_head_last(W.lm_head[lo:hi], xrow, None, W.final_norm, tc.rms_norm_eps, mine[:, :hi - lo])
        full = self._empty(R * Vl, dtype=torch.float32)
        comm.all_gather(full, mine.reshape(-1))
        return PrefillResult(logits_last=full[:tc.vocab_size], seq_len=S, n_tiles=n_tiles)

    def _llm_layers_tp(self, xs, S: int, cache: KVCache):
        """The sequence-parallel layer loop over the row chunks ``xs`` (fp32 [Sl, D] each, updated in place)."""
        ops, W, tc, comm = self.ops, self.W, self.cfg.text_config, self.comm
        R, NC, Sl, Sc = self._tp_geometry(S)
        (H, KV), hd, D_ = self._llm_heads(), tc.head_dim, tc.hidden_size
        qw, kw = H * hd, KV * hd
        dev, T = self.device, self.dtype
        cdt = self.tp_comm_dtype or T
        on_gpu = dev.type == "cuda" and not ops.emulated
        if on_gpu and self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=dev)
        cs = self._comm_stream if on_gpu else None
        ms = torch.cuda.current_stream(dev) if on_gpu else None

        def to_comm():                 # everything enqueued on the compute stream so far happens-before what follows on the comm stream
            if on_gpu:
                cs.wait_stream(ms)

        def done():                    # marker on the comm stream for the compute stream to wait on
            if not on_gpu:
                return None
            e = torch.cuda.Event()
            e.record(cs)
            return e

        def wait(e):
            if e is not None:
                ms.wait_event(e)
        pos = torch.arange(NC * Sc)
        cos, sin = self.rope_tables(pos)
        lo4 = self.lo4
        if lo4:
            # precision "lo4" under tensor parallelism: the rank's normalised rows are Lo4 pairs (16-bit rows + fp4 image of their rounding
            # residual + block scales; image and scales in ONE buffer so that one more all-gather moves both: + 27 % all-gather bytes), the
            # column-parallel GEMMs run the correction phase on the gathered pair against the image of THIS rank's weight shard, the
            # attention and the SwiGLU epilogue hand their images to the row-parallel GEMMs.  The partial products are exchanged as before.
            from .ops import lo4_packed_act
            if not (hd == 128 and W.llm_layers and W.llm_layers[0].qkv_w_rope is not None):
                raise RuntimeError("precision 'lo4' needs head_dim 128 and the rope-ordered q|k|v weights")
            L4 = self._lo4_weights("llm")
            h_loc = [lo4_packed_act(Sl, D_, T, dev) for _ in range(NC)]
            h_full = [lo4_packed_act(Sc, D_, T, dev) for _ in range(NC)]
            att, gu = self._lo4_act(Sc, qw, heads=(H, hd)), self._lo4_act(Sc, W.llm_ff)
        else:
            h_loc = [self._empty(Sl, D_) for _ in range(NC)]
            h_full = [self._empty(Sc, D_) for _ in range(NC)]
            att = self._empty(Sc, qw)
            gu = self._empty(Sc, W.llm_ff)
        part = [self._empty(Sc, D_, dtype=cdt) for _ in range(NC)]
        red = [self._empty(Sl, D_, dtype=cdt) for _ in range(NC)]
        qkv = self._empty(Sc, qw + 2 * kw)
        cu_q = torch.tensor([0, Sc], dtype=torch.int32, device=dev)
        cu_k = [torch.tensor([0, (c + 1) * Sc], dtype=torch.int32, device=dev) for c in range(NC)]
        scale = hd ** -0.5
        epi_part = _lib.EPI_STORE if cdt == T else _lib.EPI_STORE_F32
        fused_rope = self.fuse_norm_rope and hd == 128 and W.llm_layers and W.llm_layers[0].qkv_w_rope is not None
        n_layers = len(W.llm_layers)
        ev = [None] * NC

        def gather(c):
            """all-gather of chunk c's normalised rows (and, under lo4, of their residual images) on the comm stream."""
            to_comm()
            if lo4:
                comm.all_gather(h_full[c].hi, h_loc[c].hi, cs)
                comm.all_gather(h_full[c].buf.view(torch.float16), h_loc[c].buf.view(torch.float16), cs)
            else:
                comm.all_gather(h_full[c], h_loc[c], cs)
            ev[c] = done()

        def add_norm(c, delta, w):
            if w is None:
                ops.add_rmsnorm(xs[c], delta, None, None, tc.rms_norm_eps)
            elif lo4:
                ops.add_rmsnorm_lo4(xs[c], delta, w, h_loc[c], tc.rms_norm_eps)
            else:
                ops.add_rmsnorm(xs[c], delta, w, h_loc[c], tc.rms_norm_eps)
        for li, L in enumerate(W.llm_layers):
            w4 = L4[li] if lo4 else None
            # ---- attention half ----------------------------------------------------------------------------------------
            for c in range(NC):
                if li == 0:
                    if lo4:
                        ops.norm_lo4(xs[c], L.in_norm, None, h_loc[c], tc.rms_norm_eps)
                    else:
                        ops.rmsnorm(xs[c], L.in_norm, h_loc[c], tc.rms_norm_eps)
                gather(c)
            for c in range(NC):
                wait(ev[c])
                kc, vc = cache.k[li], cache.v[li]
                if lo4:
                    ops.rmsnorm_rope_lo4(h_full[c], L.qkv_w_rope, w4[0], qkv, None, tc.rms_norm_eps, cos[c * Sc:(c + 1) * Sc],
                                         sin[c * Sc:(c + 1) * Sc], kc, vc, c * Sc, H, KV, hd)
                    ops.attention_lo4(qkv[:, :qw], kc[:(c + 1) * Sc], vc[:(c + 1) * Sc], att, cu_q, cu_k[c], Sc, H, KV, hd, scale, True,
                                      window=tc.sliding_window or 0)
                    ops.gemm_lo4(att, L.o_w, w4[1], part[c], epilogue=epi_part)
                else:
                    if fused_rope:
                        ops.rmsnorm_rope(h_full[c], L.qkv_w_rope, qkv, None, tc.rms_norm_eps, cos[c * Sc:(c + 1) * Sc], sin[c * Sc:(c + 1) * Sc],
                                         kc, vc, c * Sc, H, KV, hd)
                    else:
                        ops.gemm(h_full[c], L.qkv_w, qkv)
                        ops.rope_qk(qkv, H, KV, hd, cos[c * Sc:(c + 1) * Sc], sin[c * Sc:(c + 1) * Sc], kc, vc, c * Sc)
                    ops.attention(qkv[:, :qw], kc[:(c + 1) * Sc], vc[:(c + 1) * Sc], att, cu_q, cu_k[c], Sc, H, KV, hd, scale, True,
                                  self.use_tr, window=tc.sliding_window or 0)
                    ops.gemm(att, L.o_w, part[c], epilogue=epi_part)
                to_comm()
                comm.reduce_scatter(red[c], part[c], cs)
                ev[c] = done()
            # ---- MLP half ----------------------------------------------------------------------------------------------
            for c in range(NC):
                wait(ev[c])
                add_norm(c, red[c], L.post_norm)
                gather(c)
            for c in range(NC):
                wait(ev[c])
                if lo4:
                    ops.gemm_lo4(h_full[c], L.gu_w, w4[2], gu.hi, epilogue=_lib.EPI_SWIGLU, out4=gu)
                    ops.gemm_lo4(gu, L.down_w, w4[3], part[c], epilogue=epi_part)
                else:
                    ops.gemm(h_full[c], L.gu_w, gu, epilogue=_lib.EPI_SWIGLU)
                    ops.gemm(gu, L.down_w, part[c], epilogue=epi_part)
                to_comm()
                comm.reduce_scatter(red[c], part[c], cs)
                ev[c] = done()
            for c in range(NC):
                wait(ev[c])
                add_norm(c, red[c], W.llm_layers[li + 1].in_norm if li + 1 < n_layers else None)
            if self.trace:
                self.trace(f"llm.{li}", xs)
        if on_gpu:
            ms.wait_stream(cs)

    @torch.no_grad()
    def prefill_batch(self, samples: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]]):
        """Several samples in ONE pass (BASELINE config C5: a batch of multi-image samples): all ViT inputs go through the
        vision tower and the projector together, every sample's merged sequence is packed into one varlen causal batch
        (``cu_seqlens`` keeps the samples apart).  samples: [(input_ids [1, S_in], tiles u8 [N_i, S, S, 3] or None)].
        Returns (logits_last [n_samples, vocab], seq_lens).  The reference runs one sample per ``generate`` call
        (EVAL:448-454); results are identical to per-sample ``prefill`` calls."""
        tiles = [t for _, t in samples if t is not None and t.shape[0] > 0]
        visual = None
        if tiles:
            all_tiles = torch.cat(tiles, dim=0)
            with_tiles = [(ids, t) for ids, t in samples if t is not None and t.shape[0] > 0]
            vflags = self.lo4_vit_tiles([t.shape[0] for _, t in with_tiles],
                                        [ids.numel() + t.shape[0] * (self.cfg.tokens_per_tile - 1) for ids, t in with_tiles])
            visual = self.encode_images(all_tiles, vflags)         # the graph-captured encode when `graph_encode` is set
        xs, seq_lens, row = [], [], 0
        for ids, t in samples:
            n = 0 if t is None else t.shape[0]
            vt = None if n == 0 else visual[row * self.cfg.tokens_per_tile:(row + n) * self.cfg.tokens_per_tile]
            row += n
            x = self.embed_merge(ids, vt)
            xs.append(x)
            seq_lens.append(x.shape[0])
        x = torch.cat(xs, dim=0)
        last, _ = self.llm_prefill(x, seq_lens)
        return last, seq_lens

    # ------------------------------------------------------------------------------------------------
    # a12: decode.  One step = ~290 launches of memory-bound kernels, so the step is captured once per KV cache into a
    # HIP graph over static buffers; the token id and the position live in device memory (lmi_rope_qk_at, device
    # cu_seqlens), the graph itself takes the argmax and advances the position, and the host only reads the new token.
    # ------------------------------------------------------------------------------------------------
    def _decode_state(self, cache: KVCache):
        st = getattr(cache, "_decode_state", None)
        if st is not None and getattr(st, "mode", None) == (self.precision, self.decode_precision):
            return st                                                 # (a state built under another precision mode is rebuilt: other buffers, other launches)
        W, tc = self.W, self.cfg.text_config
        (H, KV), hd, D = self._llm_heads(), tc.head_dim, tc.hidden_size
        dev = self.device

        class St:
            pass
        st = St()
        st.tok = torch.zeros(1, dtype=torch.int64, device=dev)
        st.src0 = torch.zeros(1, dtype=torch.int64, device=dev)
        st.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        st.cu_q = torch.tensor([0, 1], dtype=torch.int32, device=dev)
        st.cu_k = torch.tensor([0, 1], dtype=torch.int32, device=dev)
        st.x = self._empty(1, D, dtype=torch.float32)
        st.hl = self.decode_hl(1)                                     # operand buffers hold [hi row; lo row] pairs (decode precision mode)
        R = 2 if st.hl else 1
        st.h, st.qkv, st.att = self._empty(R, D), self._empty(1, (H + 2 * KV) * hd), self._empty(R, H * hd)
        st.gu = self._empty(R, W.llm_ff)
        st.hf32 = self._empty(1, D, dtype=torch.float32) if st.hl else None
        st.part = torch.zeros(D, dtype=torch.float32, device=dev)     # tensor parallel: partial o_proj / down_proj row
        st.sq_a, st.sq_b = self._empty(1, max(D // 16, 1), dtype=torch.float32), self._empty(1, max(D // 16, 1), dtype=torch.float32)
        st.k_begin = torch.zeros(1, dtype=torch.int32, device=dev)    # packed weights: the step runs on the batched-decode kernels with one row
        st.logits = self._empty(W.lm_head.shape[0], dtype=torch.float32)
        st.cos, st.sin = self.rope_tables(torch.arange(cache.capacity))
        st.ws = torch.empty(self.ops.decode_workspace_elems(1, H, hd, cache.capacity), dtype=torch.float32, device=dev)
        st.graph = None
        st.layout = self.llm_packed                                   # a captured step replays the launches of the weight layout it was captured on
        st.mode = (self.precision, self.decode_precision)             # ... and of the precision mode
        cache._decode_state = st
        return st

    def _decode_body(self, st, cache: KVCache):
        """Everything of one decode step that does not depend on host values."""
        ops, W, tc = self.ops, self.W, self.cfg.text_config
        (H, KV), hd = self._llm_heads(), tc.head_dim
        qw = H * hd
        tp = self.tp_size > 1

        def row_parallel(w, a):
            if not tp:
                ops.gemv(w, a, st.x[0], epilogue=2)
                return
            ops.gemv(w, a, st.part, epilogue=0)
            self.comm.all_reduce(st.part)
            st.x[0].add_(st.part)
        ops.embed_merge(st.tok, st.src0, W.embed, None, st.x)
        if self.llm_packed:
            # one copy of the weights, in the operand order of lmi_gemm_skinny: the batch-1 step is the batched step with one row (measured:
            # 3.10 ms per step at B = 2 against 3.12 ms for the GEMV step), its KV rows go to this cache, the head keeps the fp32 row
            self._skinny_layers(st, cache.k, cache.v, cache.capacity,
                                lambda i: ops.attention_decode(st.qkv[:, :qw], cache.k[i], cache.v[i], st.att, st.cu_q, st.cu_k, 1, cache.capacity, H, KV,
                                                               hd, hd ** -0.5, st.ws, window=tc.sliding_window or 0, hl=st.hl))
            ops.lm_head_last(W.lm_head, st.x, None, W.final_norm, tc.rms_norm_eps, st.logits.view(1, -1))
            ops.decode_advance(st.logits.view(1, -1), tc.vocab_size, st.tok, st.pos, k_len=st.cu_k[1:], suppress=self.suppress_tokens)
            return
        fuse = tc.hidden_size == 4096                 # lmi_gemv_rmsnorm: the norm rides in the projection's launch
        for i, L in enumerate(W.llm_layers):
            if fuse and hd == 128 and L.qkv_w_rope is not None:   # norm + projection + RoPE + KV append: one launch
                ops.gemv_rmsnorm_rope(L.qkv_w_rope, st.x[0], L.in_norm, tc.rms_norm_eps, st.qkv[0], H, KV, hd, st.cos, st.sin,
                                      cache.k[i], cache.v[i], st.pos)
            else:
                if fuse:
                    ops.gemv_rmsnorm(L.qkv_w, st.x[0], L.in_norm, tc.rms_norm_eps, st.qkv[0], epilogue=1)
                else:
                    ops.rmsnorm(st.x, L.in_norm, st.h, tc.rms_norm_eps)
                    ops.gemv(L.qkv_w, st.h[0], st.qkv[0], epilogue=1)
                ops.rope_qk_at(st.qkv, H, KV, hd, st.cos, st.sin, cache.k[i], cache.v[i], st.pos)
            ops.attention_decode(st.qkv[:, :qw], cache.k[i], cache.v[i], st.att, st.cu_q, st.cu_k, 1, cache.capacity, H, KV, hd,
                                 hd ** -0.5, st.ws, window=tc.sliding_window or 0)
            row_parallel(L.o_w, st.att[0])
            if fuse:
                ops.gemv_rmsnorm(L.gu_w, st.x[0], L.post_norm, tc.rms_norm_eps, st.gu[0], epilogue=3)
            else:
                ops.rmsnorm(st.x, L.post_norm, st.h, tc.rms_norm_eps)
                ops.gemv(L.gu_w, st.h[0], st.gu[0], epilogue=3)
            row_parallel(L.down_w, st.gu[0])
        ops.lm_head_last(W.lm_head, st.x, None, W.final_norm, tc.rms_norm_eps, st.logits.view(1, -1))
        # greedy choice and position / key-count advance stay on the device, one launch (lmi_decode_advance; capturable)
        ops.decode_advance(st.logits.view(1, -1), tc.vocab_size, st.tok, st.pos, k_len=st.cu_k[1:], suppress=self.suppress_tokens)

    def _decode_run(self, st, cache: KVCache):
        # Tensor parallel: the step holds 2 all-reduces per layer.  Through RcclComm they are plain stream-ordered RCCL launches, which
        # HIP graph capture records like any kernel (RCCL supports capture; tests/test_gpu_dist.py captures and replays
        # lmi_allreduce on the device), so the step stays ONE graph replay per token; over a torch.distributed group (gloo in the
        # CPU tests, host-staged) it cannot be captured and runs eagerly.
        from .dist import RcclComm
        if getattr(st, "layout", self.llm_packed) != self.llm_packed:      # pack_llm_weights / unpack_llm_weights since the capture
            st.graph, st.layout = None, self.llm_packed
        tp_capturable = self.tp_size == 1 or (isinstance(self.comm, RcclComm) and self.tp_decode_graph)
        if self.ops.emulated or self.device.type != "cuda" or not self.use_graphs or not tp_capturable or getattr(st, "graph_failed", False):
            self._decode_body(st, cache)
            return
        if st.graph is None:
            # warm-up outside capture (first-use attribute calls, allocator), on copies of the device counters
            keep = (st.tok.clone(), st.pos.clone(), st.cu_k.clone())
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self._decode_body(st, cache)
            torch.cuda.current_stream(self.device).wait_stream(side)
            st.tok.copy_(keep[0]); st.pos.copy_(keep[1]); st.cu_k.copy_(keep[2])
            sent0 = self.comm.sent_bytes if self.comm is not None else 0
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._decode_body(st, cache)
            except Exception as exc:                   # a communicator that cannot be captured: run the step eagerly from now on
                if self.tp_size == 1:
                    raise
                import warnings
                warnings.warn(f"tensor-parallel decode step could not be captured in a HIP graph ({exc}); running it eagerly")
                st.graph_failed = True
                torch.cuda.synchronize(self.device)
                st.tok.copy_(keep[0]); st.pos.copy_(keep[1]); st.cu_k.copy_(keep[2])
                self._decode_body(st, cache)
                return
            st.graph_comm_bytes = (self.comm.sent_bytes - sent0) if self.comm is not None else 0    # what one replay puts on the links
            if self.comm is not None:
                self.comm.sent_bytes = sent0           # capture records, it does not send
            st.tok.copy_(keep[0]); st.pos.copy_(keep[1]); st.cu_k.copy_(keep[2])      # capture does not execute
            st.graph = g
        st.graph.replay()
        if self.comm is not None:
            self.comm.sent_bytes += getattr(st, "graph_comm_bytes", 0)

    def _decode_seed(self, st, cache: KVCache, token_id: int):
        if cache.length >= cache.capacity:
            raise RuntimeError("KV cache is full")
        st.tok.fill_(int(token_id))
        st.pos.fill_(cache.length)
        st.cu_k[1:].fill_(cache.length + 1)

    @torch.no_grad()
    def decode_step(self, token_id: int, cache: KVCache) -> torch.Tensor:
        """Append one token: returns its logits [vocab] (fp32, a view of a static buffer) and advances the cache."""
        st = self._decode_state(cache)
        self._decode_seed(st, cache, token_id)
        self._decode_run(st, cache)
        cache.length += 1
        return st.logits[:self.cfg.text_config.vocab_size]

    def _generation_cache(self, need: int) -> KVCache:
        """ONE KV cache (with its decode state and captured graph) per engine, reused across generate() calls: the eval loop is
        batch 1 with at most 128 new tokens, and a fresh cache per sample would pay an eager warm-up step, a graph capture and new
        static buffers every time.  The graph reads the position and the key count from device memory and its launch geometry
        depends only on the capacity, so resetting ``length`` is all a new prompt needs; a longer prompt grows the cache."""
        c = getattr(self, "_gen_cache", None)
        if c is None or c.capacity < need:
            c = self._gen_cache = KVCache(self.cfg, (need + 2047) // 2048 * 2048, self.dtype, self.device, tp_size=self.tp_size)
        c.length = 0
        return c

    def first_token(self, logits_last: torch.Tensor) -> int:
        """Greedy choice from prefill logits, honouring ``suppress_tokens``."""
        if self.suppress_tokens is not None:
            logits_last = logits_last.clone()
            logits_last.index_fill_(0, self.suppress_tokens, float("-inf"))
        return int(logits_last.argmax())

    def _greedy_loop(self, prompt_ids: List[int], first: int, cache: KVCache, max_new_tokens: int, eos) -> List[int]:
        """EVAL:448-452 after the prefill: greedy tokens until eos / max_new_tokens; one captured decode step per token."""
        out, nxt = list(prompt_ids), int(first)
        st = self._decode_state(cache)
        self._decode_seed(st, cache, nxt)
        for step in range(max_new_tokens):
            out.append(nxt)
            if nxt in eos or step == max_new_tokens - 1:
                break
            self._decode_run(st, cache)              # consumes st.tok at st.pos, leaves the next token / position on the device
            cache.length += 1
            nxt = int(st.tok.item())
        return out

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, tiles: Optional[torch.Tensor], max_new_tokens: int = 128,
                 eos_token_id: Sequence[int] = (128001, 128009)) -> torch.Tensor:
        """Greedy generation (EVAL:448-452): returns LongTensor [1, S_in + T] on the input device."""
        ids = input_ids.reshape(1, -1)
        n_img = int((ids == self.cfg.image_token_index).sum())
        S = ids.shape[1] + n_img * (self.cfg.tokens_per_tile - 1)
        cache = self._generation_cache((self.tp_padded_len(S) if self.tp_size > 1 else S) + max_new_tokens)
        res = self.prefill(ids, tiles, cache=cache)
        out = self._greedy_loop([int(t) for t in ids.reshape(-1).tolist()], self.first_token(res.logits_last), cache, max_new_tokens,
                                set(int(e) for e in eos_token_id))
        return torch.tensor([out], dtype=torch.long, device=input_ids.device)

    # ------------------------------------------------------------------------------------------------
    # f4: batched decode.  B sequences advance together: ONE pass over the weights per step serves B tokens (lmi_gemm_skinny),
    # their KV caches are slots of one pooled buffer per layer (slot b = rows [b * capacity, (b + 1) * capacity)), positions and
    # key counts live in device memory, and the whole step is ONE captured HIP graph per (B, capacity).
    # ------------------------------------------------------------------------------------------------
    MAX_DECODE_BATCH = 16              # rows of one lmi_gemm_skinny launch (one 16x16x32 MFMA column block)
    HIST = 8                           # continuous batching: decode steps between two host looks at the produced tokens
    MAX_EOS = 4                        # eos ids held on the device

    def _batch_decode_supported(self) -> bool:
        """lmi_gemm_skinny needs K % 128 == 0 and N % 16 == 0 (gate/up: % 64); the pooled decode attention head_dim 128."""
        W, tc = self.W, self.cfg.text_config
        (H, KV), hd, D = self._llm_heads(), tc.head_dim, tc.hidden_size
        return (hd == 128 and D % 128 == 0 and W.llm_ff % 128 == 0 and (2 * W.llm_ff) % 64 == 0 and ((H + 2 * KV) * hd) % 16 == 0 and
                W.lm_head.shape[0] % 16 == 0 and D % 16 == 0)

    def _batch_state(self, B: int, need: int):
        """Static buffers + pooled KV cache + captured graph of a B-sequence decode step (kept per engine and reused: the graph reads
        positions / key counts from device memory and its launch geometry depends on (B, capacity) only)."""
        states = getattr(self, "_batch_states", None)
        if states is None:
            states = self._batch_states = {}
        st = states.get(B)
        if st is not None and st.capacity >= need and getattr(st, "mode", None) == (self.precision, self.decode_precision):
            return st
        W, tc, dev = self.W, self.cfg.text_config, self.device
        (H, KV), hd, D = self._llm_heads(), tc.head_dim, tc.hidden_size
        cap = (need + 1023) // 1024 * 1024

        class St:
            pass
        st = St()
        st.B, st.capacity = B, cap
        n_layers = len(W.llm_layers)
        st.k = [torch.zeros(B * cap, KV * hd, dtype=self.dtype, device=dev) for _ in range(n_layers)]
        st.v = [torch.zeros(B * cap, KV * hd, dtype=self.dtype, device=dev) for _ in range(n_layers)]
        st.tok = torch.zeros(B, dtype=torch.int64, device=dev)
        st.src = torch.arange(B, dtype=torch.int64, device=dev)
        st.pos = torch.zeros(B, dtype=torch.int32, device=dev)
        st.k_len = torch.ones(B, dtype=torch.int32, device=dev)
        st.k_begin = (torch.arange(B, dtype=torch.int32) * cap).to(dev)
        st.cu_q = torch.arange(B + 1, dtype=torch.int32, device=dev)
        st.x = self._empty(B, D, dtype=torch.float32)
        st.hl = self.decode_hl(B)                                     # [hi rows; lo rows] operand pairs (decode precision mode; 2 B <= 16)
        R = 2 * B if st.hl else B
        st.h, st.qkv, st.att = self._empty(R, D), self._empty(B, (H + 2 * KV) * hd), self._empty(R, H * hd)
        st.gu = self._empty(R, W.llm_ff)
        st.hf32 = self._empty(B, D, dtype=torch.float32) if st.hl else None
        st.mode = (self.precision, self.decode_precision)
        st.sq_a, st.sq_b = self._empty(B, D // 16, dtype=torch.float32), self._empty(B, D // 16, dtype=torch.float32)   # folded-norm partials
        st.logits = self._empty(B, W.lm_head.shape[0], dtype=torch.float32)
        st.cos, st.sin = self.rope_tables(torch.arange(cap))
        st.ws = torch.empty(self.ops.decode_workspace_elems(B, H, hd, cap), dtype=torch.float32, device=dev)
        # continuous batching (generate_stream): which slots hold a running sequence, how many more tokens each may produce, the eos ids,
        # and a ring of the last HIST steps' tokens — all in device memory, so that slots are admitted / retired between replays of ONE
        # captured step and the host looks at the tokens only every HIST steps
        st.live = torch.ones(B, dtype=torch.int32, device=dev)
        st.budget = torch.full((B,), 1 << 30, dtype=torch.int32, device=dev)
        st.eos = torch.full((self.MAX_EOS,), -1, dtype=torch.int64, device=dev)
        st.hist = torch.zeros(self.HIST, B, dtype=torch.int64, device=dev)
        st.hist_pos = torch.zeros(B, dtype=torch.int32, device=dev)
        st.graph = None
        # bounded: a serving process that sees many batch sizes keeps the pools of the two most recent ones (each is B x capacity KV rows)
        while len(states) >= 2:
            states.pop(next(iter(states)))
        states[B] = st
        return st

    def _skinny_weights(self):
        """The decode projections + the head in lmi_gemm_skinny's packed (MFMA operand) order — a second copy of the 16-bit LLM weights
        (15 GB for Llama-3.1-8B), built on the first batched decode and shared by every batch size.  Only engines whose weights stay in
        the nn.Linear layout (tensor parallel, pack_llm_weights=False) need it: the default layout IS this order (pack_llm_weights)."""
        pk = getattr(self, "_skinny_pack", None)
        if pk is None:
            from .weights import skinny_pack
            W = self.W
            # q|k|v: the rope-permuted rows when they exist (RoPE + KV append then ride in the projection's epilogue: lmi_rope_qkv_skinny)
            pk = self._skinny_pack = {"layers": [(skinny_pack(L.qkv_w_rope if L.qkv_w_rope is not None else L.qkv_w), skinny_pack(L.o_w),
                                                  skinny_pack(L.gu_w), skinny_pack(L.down_w)) for L in W.llm_layers],
                                      "head": self._skinny_head()}
        return pk

    def _skinny_layers(self, st, k_list, v_list, capacity: int, attend):
        """The layer stack of a decode step on the M <= 16 kernels (lmi_gemm_skinny*): st.x (fp32 rows) in, st.x out.  K / V rows are appended
        to k_list[i] / v_list[i] at row m * capacity + pos[m]; ``attend(i)`` runs layer i's attention from st.qkv into st.att."""
        ops, W, tc = self.ops, self.W, self.cfg.text_config
        (H, KV), hd = self._llm_heads(), tc.head_dim
        eps = tc.rms_norm_eps
        pk = None if self.llm_packed or not self.skinny_packed else self._skinny_weights()
        D, n_layers = tc.hidden_size, len(W.llm_layers)
        # folded RMSNorms (as the prefill's fused schedule): every residual projection (o_proj, down_proj) also emits T(x * gamma_next) and
        # per-row partial sums of squares, the projection that consumes them scales its accumulator rows by rstd — only the first
        # layer's norm and the final one stay launches of their own
        fold = self.skinny_fold_norm and D % 16 == 0 and all(L.qkv_w_rope is not None for L in W.llm_layers) and hd == 128
        hl = bool(getattr(st, "hl", False))                           # decode precision mode: operands are [hi rows; lo rows] pairs (decode_hl)
        assert not hl or fold
        for i, L in enumerate(W.llm_layers):
            rope_fused = L.qkv_w_rope is not None and hd == 128
            qkv_w, o_w, gu_w, down_w = pk["layers"][i] if pk else (L.qkv_w_rope if rope_fused else L.qkv_w, L.o_w, L.gu_w, L.down_w)
            packed = None if pk is None else True            # None: as the weight is marked (weights.is_packed)
            if hl and i == 0:
                ops.rmsnorm(st.x, L.in_norm, st.hf32, eps)            # the first norm in fp32, handed over as a pair
                ops.split_rows_hl(st.hf32, st.h)
            elif not fold or i == 0:
                ops.rmsnorm(st.x, L.in_norm, st.h, eps)
            if rope_fused:
                ops.rope_qkv_skinny(qkv_w, st.h, st.qkv, H, KV, hd, st.cos, st.sin, k_list[i], v_list[i], capacity, st.pos, packed,
                                    rowsq_in=st.sq_b if fold and i > 0 else None, norm_eps=eps, hl=hl)
            else:
                ops.gemm_skinny(qkv_w, st.h, st.qkv, 0, packed)
                ops.rope_qk_rows(st.qkv, H, KV, hd, st.cos, st.sin, k_list[i], v_list[i], capacity, st.pos)
            attend(i)
            if fold:
                ops.gemm_skinny(o_w, st.att, st.x, 1, packed, norm_out=st.h, norm_gamma=L.post_norm, rowsq_out=st.sq_a, hl=hl)
                ops.gemm_skinny(gu_w, st.h, st.gu, 2, packed, rowsq_in=st.sq_a, norm_dim=D, norm_eps=eps, hl=hl)
                if i + 1 < n_layers:
                    ops.gemm_skinny(down_w, st.gu, st.x, 1, packed, norm_out=st.h, norm_gamma=W.llm_layers[i + 1].in_norm, rowsq_out=st.sq_b, hl=hl)
                else:
                    ops.gemm_skinny(down_w, st.gu, st.x, 1, packed, hl=hl)
            else:
                ops.gemm_skinny(o_w, st.att, st.x, 1, packed)
                ops.rmsnorm(st.x, L.post_norm, st.h, eps)
                ops.gemm_skinny(gu_w, st.h, st.gu, 2, packed)
                ops.gemm_skinny(down_w, st.gu, st.x, 1, packed)

    def _skinny_head(self):
        """lm_head in the packed order for the batched step (shared with _skinny_weights' second copy when there is one)."""
        h = getattr(self, "_head_pack", None)
        if h is None:
            from .weights import as_packed
            pk = getattr(self, "_skinny_pack", None)
            h = self._head_pack = as_packed(pk["head"]) if pk else as_packed(self.W.lm_head)
        return h

    def _batch_decode_body(self, st):
        """One decode step for the B sequences of ``st`` (everything here is host-value free: graph-capturable)."""
        ops, W, tc = self.ops, self.W, self.cfg.text_config
        (H, KV), hd = self._llm_heads(), tc.head_dim
        qw, eps = H * hd, tc.rms_norm_eps
        ops.embed_merge(st.tok, st.src, W.embed, None, st.x)
        self._skinny_layers(st, st.k, st.v, st.capacity,
                            lambda i: ops.attention_decode_pool(st.qkv[:, :qw], st.k[i], st.v[i], st.att, st.cu_q, st.k_begin, st.k_len, st.capacity,
                                                                H, KV, hd, hd ** -0.5, st.ws, window=tc.sliding_window or 0, hl=st.hl))
        # head: one pass over lm_head for all B rows (lmi_lm_head_last streams the 1 GB head once PER row), from a packed copy of the head
        # (1 GB, built on the first batched step: the prefill's and the batch-1 step's head kernels read the nn.Linear layout; B = 8 step
        # 3.29 ms against 3.37 ms from the row-major head in the coalescing lane order)
        if st.hl:                                                     # decode precision mode: the head's operand as a pair too (the batch-1 step's head reads the fp32 row)
            ops.rmsnorm(st.x, W.final_norm, st.hf32, eps)
            ops.split_rows_hl(st.hf32, st.h)
            ops.gemm_skinny(self._skinny_head() if self.skinny_packed else W.lm_head, st.h, st.logits, 3, hl=True)
        else:
            ops.rmsnorm(st.x, W.final_norm, st.h, eps)
            ops.gemm_skinny(self._skinny_head() if self.skinny_packed else W.lm_head, st.h, st.logits, 3)
        # greedy choice, history ring, stop rule (eos ids / token budget) and position advance of all B slots: ONE launch, device memory only —
        # a slot that stopped freezes (live = 0) and what it produces afterwards is ignored (lmi_decode_advance)
        ops.decode_advance(st.logits, tc.vocab_size, st.tok, st.pos, k_len=st.k_len, live=st.live, budget=st.budget, eos=st.eos, hist=st.hist,
                           hist_pos=st.hist_pos, suppress=self.suppress_tokens)

    def _batch_decode_run(self, st):
        if self.ops.emulated or self.device.type != "cuda" or not self.use_graphs:
            self._batch_decode_body(st)
            return
        if st.graph is None:
            names = ("tok", "pos", "k_len", "live", "budget", "hist", "hist_pos")
            keep = [getattr(st, n).clone() for n in names]
            side = torch.cuda.Stream(device=self.device)               # warm-up outside capture (function attributes, allocator)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self._batch_decode_body(st)
            torch.cuda.current_stream(self.device).wait_stream(side)
            for n, v in zip(names, keep):
                getattr(st, n).copy_(v)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._batch_decode_body(st)
            for n, v in zip(names, keep):                              # capture does not execute
                getattr(st, n).copy_(v)
            st.graph = g
        st.graph.replay()

    def _greedy_loop_batch(self, st, prompts: List[List[int]], first: List[int], seq_lens: List[int], max_new_tokens: int, eos) -> List[List[int]]:
        """EVAL:448-452 for B sequences at once: every sequence follows exactly the batch-1 rule (emit, stop at eos / max_new_tokens);
        finished sequences keep riding along in the batch (their slots are private) and are ignored."""
        B = st.B
        outs = [list(p) for p in prompts]
        nxt = [int(f) for f in first]
        done = [False] * B
        st.tok.copy_(torch.tensor(nxt, dtype=torch.int64))
        st.pos.copy_(torch.tensor(seq_lens, dtype=torch.int32))
        st.k_len.copy_(torch.tensor([s + 1 for s in seq_lens], dtype=torch.int32))
        st.live.fill_(1); st.budget.fill_(1 << 30); st.eos.fill_(-1); st.hist_pos.zero_()      # the host applies the stop rule here
        for step in range(max_new_tokens):
            for j in range(B):
                if not done[j]:
                    outs[j].append(nxt[j])
                    if nxt[j] in eos or step == max_new_tokens - 1:
                        done[j] = True
            if all(done):
                break
            self._batch_decode_run(st)
            nxt = [int(t) for t in st.tok.tolist()]
        return outs

    def release_batch_state(self) -> None:
        """Free the pooled KV caches / captured graphs of the batched decode and — for engines that keep the nn.Linear layout (tensor
        parallel, pack_llm_weights=False) — the second copy of the LLM weights in the skinny-M operand order (15 GB for Llama-3.1-8B;
        rebuilt on the next batched call).  The default layout has no second copy: prefill, batch-1 and batched decode read the same
        packed tensors (pack_llm_weights)."""
        self._batch_states = {}
        self._skinny_pack = None
        self._head_pack = None
        if self.device.type == "cuda":
            torch.cuda.empty_cache()

    @torch.no_grad()
    def generate_stream(self, samples: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]], batch_size: int = 8, max_new_tokens: int = 128,
                        eos_token_id: Sequence[int] = (128001, 128009), stats: Optional[dict] = None) -> List[torch.Tensor]:
        """CONTINUOUS batching (the reference loop EVAL:381-452 is one generate() per record): ``batch_size`` decode slots,
        ONE captured step per token for all of them, and a slot that finishes (eos / max_new_tokens) is handed to the next pending sample
        — prefill of the newcomer, its K / V rows copied into the slot of the pooled cache, five small device writes — without
        re-capturing anything: positions, key counts, the live mask, the token budget and the eos ids all live in device memory, and the
        stop rule runs on the device.  The host reads the produced tokens every HIST steps (one copy of a [HIST, B] table) instead of
        one blocking read per token; a slot that stopped inside the window idles until the window ends (its state is frozen).
        Returns the outputs in input order, each exactly what ``generate`` returns for that sample (same rule; the batched projections sum
        in a different order than the batch-1 GEMVs, so a token can differ only on a near tie of the top two logits).
        ``stats`` (optional dict) receives steps / slot-steps / live slot-steps for occupancy accounting."""
        assert self.tp_size == 1, "batched generation is a single-rank feature (replicas scale it out)"
        B = max(1, min(int(batch_size), self.MAX_DECODE_BATCH, len(samples)))
        eos = [int(e) for e in eos_token_id]
        if B == 1 or not self._batch_decode_supported() or len(eos) > self.MAX_EOS:
            return [self.generate(ids, t() if callable(t) else t, max_new_tokens, eos) for ids, t in samples]
        tpt = self.cfg.tokens_per_tile
        def merged_len(ids):
            return ids.shape[-1] + int((ids == self.cfg.image_token_index).sum()) * (tpt - 1)
        need = max(merged_len(ids) for ids, _ in samples) + max_new_tokens
        st = self._batch_state(B, need)
        st.eos.fill_(-1)
        if eos:
            st.eos[:len(eos)].copy_(torch.tensor(eos, dtype=torch.int64))
        st.live.zero_(); st.budget.zero_(); st.pos.zero_(); st.k_len.fill_(1); st.tok.zero_(); st.hist_pos.zero_()
        eos_set = set(eos)
        outs: List[Optional[List[int]]] = [None] * len(samples)
        slot_sample = [-1] * B                                       # which sample a slot runs (-1: free)
        h_budget = [0] * B                                           # host mirror of the device stop rule
        pending = list(range(len(samples)))
        n_steps = slot_steps = live_steps = 0
        scratch = getattr(self, "_stream_cache", None)

        def admit(j: int) -> bool:
            """Next pending sample into slot j; False when nothing is pending.  Samples that end with their first token never take a slot."""
            nonlocal scratch
            while pending:
                i = pending.pop(0)
                ids, tiles = samples[i]
                if callable(tiles):                                   # lazy pixels: prepared when the sample is admitted, dropped after its prefill
                    tiles = tiles()
                S = merged_len(ids)
                if scratch is None or scratch.capacity < S:
                    scratch = self._stream_cache = KVCache(self.cfg, (S + 1023) // 1024 * 1024, self.dtype, self.device)
                scratch.length = 0
                res = self.prefill(ids.reshape(1, -1), tiles, cache=scratch)
                first = self.first_token(res.logits_last)
                out = [int(t) for t in ids.reshape(-1).tolist()] + [first]
                outs[i] = out
                if first in eos_set or max_new_tokens <= 1:
                    continue                                          # finished by the prefill alone
                for li in range(len(scratch.k)):
                    self.ops.kv_append(scratch.k[li][:S], scratch.v[li][:S], st.k[li], st.v[li], j * st.capacity)
                dev = self.device
                st.tok[j:j + 1].copy_(torch.tensor([first], dtype=torch.int64, device=dev))
                st.pos[j:j + 1].copy_(torch.tensor([S], dtype=torch.int32, device=dev))
                st.k_len[j:j + 1].copy_(torch.tensor([S + 1], dtype=torch.int32, device=dev))
                st.budget[j:j + 1].copy_(torch.tensor([max_new_tokens - 1], dtype=torch.int32, device=dev))
                st.live[j:j + 1].fill_(1)
                slot_sample[j], h_budget[j] = i, max_new_tokens - 1
                return True
            return False

        for j in range(B):
            admit(j)
        while any(i >= 0 for i in slot_sample):
            window = self.HIST
            st.hist_pos.zero_()
            for _ in range(window):
                self._batch_decode_run(st)
            toks = st.hist.tolist()                                   # ONE host read per window: [HIST][B]
            n_steps += window
            for w in range(window):
                for j in range(B):
                    i = slot_sample[j]
                    slot_steps += 1
                    if i < 0:
                        continue
                    live_steps += 1
                    t = int(toks[w][j])
                    outs[i].append(t)
                    h_budget[j] -= 1
                    if t in eos_set or h_budget[j] <= 0:
                        slot_sample[j] = -1                           # retired: the device froze it at this very step
            for j in range(B):
                if slot_sample[j] < 0:
                    admit(j)
        if stats is not None:
            stats.update(steps=n_steps, slot_steps=slot_steps, live_slot_steps=live_steps, batch_size=B)
        return [torch.tensor([o], dtype=torch.long, device=samples[i][0].device) for i, o in enumerate(outs)]

    @torch.no_grad()
    def generate_batch(self, samples: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]], max_new_tokens: int = 128,
                       eos_token_id: Sequence[int] = (128001, 128009)) -> List[torch.Tensor]:
        """Several samples per call (batching with per-sample cu_seqlens instead of one sample per generate()):
        ONE packed prefill — all ViT inputs through the tower together, all merged sequences in one varlen causal pass that also
        writes every sample's K/V into a packed cache — then each sample's K/V rows move to its slot of the pooled decode cache (a
        device copy) and ALL samples continue greedily together, one captured decode step per token for the whole batch
        (``_batch_decode_body``: the weight stream of a step is shared by the batch).  Same RULE as per-sample ``generate`` and numerically
        equivalent, not bit-identical: the first new token comes from the same prefill, the continuation's projections are MFMA tiles with folded
        norms instead of the batch-1 FMA chains, so a greedy choice can differ where the top two logits are within the 16-bit noise (the GPU
        tests assert equality and, where it fails, exactly such a near tie)."""
        assert self.tp_size == 1, "batched generation is a single-rank feature (replicas scale it out)"
        if len(samples) > self.MAX_DECODE_BATCH:
            outs = []
            for i in range(0, len(samples), self.MAX_DECODE_BATCH):
                outs += self.generate_batch(samples[i:i + self.MAX_DECODE_BATCH], max_new_tokens, eos_token_id)
            return outs
        if len(samples) == 1:
            ids, t = samples[0]
            return [self.generate(ids, t, max_new_tokens, eos_token_id)]
        tiles = [t for _, t in samples if t is not None and t.shape[0] > 0]
        visual = None
        if tiles:
            all_tiles = torch.cat(tiles, dim=0)
            with_tiles = [(ids, t) for ids, t in samples if t is not None and t.shape[0] > 0]
            visual = self.encode_images(all_tiles, self.lo4_vit_tiles([t.shape[0] for _, t in with_tiles],
                                                                      [ids.numel() + t.shape[0] * (self.cfg.tokens_per_tile - 1) for ids, t in with_tiles]))
        xs, seq_lens, row = [], [], 0
        for ids, t in samples:
            n = 0 if t is None else t.shape[0]
            vt = None if n == 0 else visual[row * self.cfg.tokens_per_tile:(row + n) * self.cfg.tokens_per_tile]
            row += n
            x = self.embed_merge(ids, vt)
            xs.append(x)
            seq_lens.append(x.shape[0])
        packed = KVCache(self.cfg, sum(seq_lens), self.dtype, self.device)
        last, _ = self.llm_prefill(torch.cat(xs, dim=0), seq_lens, cache=packed)
        first = [self.first_token(last[j]) for j in range(last.shape[0])]
        eos = set(int(e) for e in eos_token_id)
        if not self._batch_decode_supported():
            # shapes the skinny-M kernels do not cover (toy configurations): the samples continue one after another on the engine's ONE
            # generation cache and its captured batch-1 step
            outs, off = [], 0
            for (ids, _), S, nxt in zip(samples, seq_lens, first):
                cache = self._generation_cache(S + max_new_tokens)
                for i in range(len(cache.k)):
                    self.ops.kv_append(packed.k[i][off:off + S], packed.v[i][off:off + S], cache.k[i], cache.v[i], 0)
                cache.length = S
                off += S
                out = self._greedy_loop([int(t) for t in ids.reshape(-1).tolist()], nxt, cache, max_new_tokens, eos)
                outs.append(torch.tensor([out], dtype=torch.long, device=ids.device))
            return outs
        st = self._batch_state(len(samples), max(seq_lens) + max_new_tokens)
        off = 0
        for j, S in enumerate(seq_lens):
            for i in range(len(packed.k)):
                self.ops.kv_append(packed.k[i][off:off + S], packed.v[i][off:off + S], st.k[i], st.v[i], j * st.capacity)
            off += S
        del packed
        prompts = [[int(t) for t in ids.reshape(-1).tolist()] for ids, _ in samples]
        outs = self._greedy_loop_batch(st, prompts, first, seq_lens, max_new_tokens, eos)
        return [torch.tensor([o], dtype=torch.long, device=ids.device) for o, (ids, _) in zip(outs, samples)]

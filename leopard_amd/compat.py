"""The model object the reference's evaluation script drives (evaluations/models/llava_multiimg_siglip_anyres.py,
"EVAL"), backed by the HIP engine.

EVAL:373-375   llava = myLlavaForConditionalGeneration.from_pretrained(ckpt, torch_dtype=torch.float32); .eval(); .to('cuda:0')
EVAL:405,445   images.to(llava.device), input_ids.to(llava.device)
EVAL:448-452   llava.generate(input_ids, pixel_values=, attention_mask=, pad_token_id=, eos_token_id=[128001,128009],
                              max_new_tokens=128, use_cache=True) -> LongTensor [1, S_in + T]
EVAL:201-361   forward(input_ids=, pixel_values=, attention_mask=, ..., return_dict=) -> .logits [1,S,V], .past_key_values

``LeopardForConditionalGeneration`` honours exactly that surface (same argument names, same return shapes/devices,
ValueError on an image-token / feature-count mismatch as transformers 4.38's merge raises).  ``torch_dtype`` is the
dtype the caller's tensors use (fp32 in EVAL); the MFMA compute type is ``compute_dtype`` (fp16 by default: closest to
the reference's fp32 results at the full matrix-core rate).  Batch is 1 per call, like the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import os

import torch

from .checkpoint import CheckpointSource, load_config
from .config import LeopardConfig
from .engine import KVCache, LeopardEngine
from .ops import Ops
from .weights import EngineWeights


@dataclass
class LlavaCausalLMOutputWithPast:
    """Field-compatible with transformers.models.llava.modeling_llava.LlavaCausalLMOutputWithPast (EVAL:355-361)."""
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[object] = None
    hidden_states: Optional[Tuple] = None
    attentions: Optional[Tuple] = None

    def __getitem__(self, i):
        return [v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions)
                if v is not None][i]


def resolve_precision(requested_dtype, compute_dtype, precision: Optional[str] = None, tp_size: int = 1) -> str:
    """Schedule of the prefill (LeopardEngine.precision).  ``LEOPARD_AMD_PRECISION`` = fast | lo4 | split overrides everything; an explicit
    ``precision`` argument comes next; otherwise a caller that asks for ``torch_dtype=torch.float32`` — the reference script does, EVAL:373 —
    gets the mode that meets the stated tolerance against fp32 arithmetic (lo4: fp16 operands + the fp4 correction phase, full-depth logits
    within 1e-3), and a caller that asks for a 16-bit type gets the fast schedule of that type.  Tensor-parallel engines run fast or lo4."""
    env = os.environ.get("LEOPARD_AMD_PRECISION", "").lower()
    mode = env or precision or ("lo4" if requested_dtype in (None, torch.float32) and compute_dtype == torch.float16 else "fast")
    if mode not in ("fast", "lo4", "split"):
        raise ValueError(f"precision must be fast, lo4 or split, not {mode!r}")
    return "fast" if (tp_size > 1 and mode == "split") else mode


class LeopardForConditionalGeneration:
    def __init__(self, config: LeopardConfig, source_factory, compute_dtype=torch.float16, ops: Optional[Ops] = None,
                 torch_dtype=torch.float32, precision: Optional[str] = None, tp_rank: int = 0, tp_size: int = 1):
        self.config = config
        self._source_factory = source_factory            # (device, dtype) -> parameter source with .get(name)
        self.compute_dtype = compute_dtype
        self.requested_dtype = torch_dtype
        self.precision = resolve_precision(torch_dtype, compute_dtype, precision, tp_size)
        self.tp_rank, self.tp_size = int(tp_rank), int(tp_size)
        self._ops = ops
        self._engine: Optional[LeopardEngine] = None
        self.device = torch.device("cpu")

    # ---- loading ---------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=torch.float32, compute_dtype=torch.float16, ops: Optional[Ops] = None,
                        precision: Optional[str] = None, tp_rank: int = 0, tp_size: int = 1):
        """``precision``: see resolve_precision.  ``tp_size`` > 1 (SURVEY.md 8 f1 "optional TP pre-sharding on load"): this process holds
        tensor-parallel shard ``tp_rank`` of the LLM — the checkpoint's tensors are sliced while they stream to the device
        (EngineWeights.build) — and needs an initialised process group of that size (leopard_amd.dist.init) by the time ``.to(device)`` runs."""
        cfg = load_config(path)
        return cls(cfg, lambda dev, dt: CheckpointSource(path, dev, dt), compute_dtype, ops, torch_dtype=torch_dtype, precision=precision,
                   tp_rank=tp_rank, tp_size=tp_size)

    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if self._engine is None or device != self.device:
            ops = self._ops if self._ops is not None else Ops()
            source = self._source_factory(device, self.compute_dtype)
            W = EngineWeights.build(self.config, source, self.compute_dtype, tp_rank=self.tp_rank, tp_size=self.tp_size)
            stats = getattr(source, "cast_stats", None)              # checkpoint.CheckpointSource: what the cast to the compute type cost
            self.weight_cast_stats = dict(stats) if stats else None
            if stats and stats["inexact_elements"] > 0.05 * max(stats["elements"], 1):     # (a bf16-trained checkpoint: only its few values below 2^-17)
                import warnings
                warnings.warn("leopard_amd: " + source.cast_report(), UserWarning, stacklevel=2)
            self._engine = LeopardEngine(self.config, W, ops=ops, device=device)
            if self.precision == "lo4" and not self._engine.lo4_supported():
                # a model shape the lo4 schedule does not cover: the 2 K mode meets the same figure on one rank; tensor-parallel engines only
                # run fast / lo4 (resolve_precision's rule for an explicit "split" request)
                self.precision = "split" if self.tp_size == 1 else "fast"
            self._engine.precision = self.precision
            self.device = device
        return self

    @property
    def engine(self) -> LeopardEngine:
        if self._engine is None:
            raise RuntimeError("call .to(device) first (the HIP engine is built on the target device)")
        return self._engine

    def _as_tiles(self, pixel_values):
        """``pixel_values``: the reference's normalised fp32 [N,3,S,S], or the GPU tiler's u8 [N,S,S,3] tile stack (the
        normalisation then happens in lmi_preprocess_tiles)."""
        if pixel_values is None:
            return None
        if pixel_values.dtype == torch.uint8:
            return pixel_values.to(self.device).contiguous()
        return pixel_values.to(device=self.device, dtype=torch.float32).contiguous()

    # ---- EVAL:201-361 ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_ids=None, pixel_values=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, vision_feature_layer=None, vision_feature_select_strategy=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        if inputs_embeds is not None or labels is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("inference surface only: input_ids (+ pixel_values), no labels / attentions")
        if input_ids.shape[0] != 1:
            raise NotImplementedError("batch 1 per call, as in the reference harness (EVAL:381-452)")
        if attention_mask is not None and not bool(attention_mask.to(torch.bool).all()):
            raise NotImplementedError("padded prompts are not produced by the reference harness (batch 1)")
        eng = self.engine
        if past_key_values is not None and input_ids.shape[1] == 1:          # decode branch, EVAL:291-320
            logits = eng.decode_step(int(input_ids[0, 0]), past_key_values)
            return LlavaCausalLMOutputWithPast(logits=logits.view(1, 1, -1), past_key_values=past_key_values)
        tiles = self._as_tiles(pixel_values)
        S = input_ids.shape[1] + int((input_ids == self.config.image_token_index).sum()) * (self.config.tokens_per_tile - 1)
        cache = KVCache(self.config, S + 256, self.compute_dtype, self.device) if use_cache else None
        res = eng.prefill(input_ids.to(self.device), tiles, cache=cache, all_logits=True)
        return LlavaCausalLMOutputWithPast(logits=res.logits_all.unsqueeze(0), past_key_values=cache)

    __call__ = forward

    # ---- EVAL:448-452 ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids, pixel_values=None, attention_mask=None, pad_token_id=None, eos_token_id=None,
                 max_new_tokens: int = 128, use_cache: bool = True, **unused):
        if input_ids.shape[0] != 1:
            raise NotImplementedError("batch 1 per call, as in the reference harness")
        eos = eos_token_id if isinstance(eos_token_id, (list, tuple)) else ([] if eos_token_id is None else [eos_token_id])
        tiles = self._as_tiles(pixel_values)
        return self.engine.generate(input_ids.to(self.device), tiles, max_new_tokens=max_new_tokens, eos_token_id=eos)


    @torch.no_grad()
    def generate_batch(self, requests: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]], eos_token_id=None,
                       max_new_tokens: int = 128, **unused) -> List[torch.Tensor]:
        """[(input_ids [1, S_in], pixel_values)] -> [LongTensor [1, S_in + T]]: the batched form of ``generate`` (one packed prefill
        for all requests, LeopardEngine.generate_batch).  Not a surface of the reference script — its loop is batch 1 — but of
        leopard_amd.harness.run_inference(batch_size=...)."""
        eos = eos_token_id if isinstance(eos_token_id, (list, tuple)) else ([] if eos_token_id is None else [eos_token_id])
        samples = [(ids.to(self.device), self._as_tiles(pix)) for ids, pix in requests]
        return self.engine.generate_batch(samples, max_new_tokens=max_new_tokens, eos_token_id=eos)


    @torch.no_grad()
    def generate_stream(self, requests: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]], batch_size: int = 8, eos_token_id=None,
                        max_new_tokens: int = 128, stats: Optional[dict] = None, **unused) -> List[torch.Tensor]:
        """Continuous batching over a list of requests: ``batch_size`` decode slots kept busy (LeopardEngine.generate_stream); the outputs
        come back in request order, each what ``generate`` returns for that request."""
        eos = eos_token_id if isinstance(eos_token_id, (list, tuple)) else ([] if eos_token_id is None else [eos_token_id])
        # a callable in place of the pixels is called when a slot admits the request (leopard_amd.harness: bounded memory over a shard)
        samples = [(ids.to(self.device), (lambda f=pix: self._as_tiles(f())) if callable(pix) else self._as_tiles(pix)) for ids, pix in requests]
        return self.engine.generate_stream(samples, batch_size=batch_size, max_new_tokens=max_new_tokens, eos_token_id=eos, stats=stats)


def from_pretrained(path: str, torch_dtype=torch.float32, **kw) -> LeopardForConditionalGeneration:
    """Drop-in for ``myLlavaForConditionalGeneration.from_pretrained``."""
    return LeopardForConditionalGeneration.from_pretrained(path, torch_dtype=torch_dtype, **kw)

"""The eval-script-facing surface of Leopard-Idefics2 (evaluations/models/idefics2_multiimg.py, "IDEF" below), on the HIP engine.

IDEF:22-30, 88-97 uses, in this order:
    processor = AutoProcessor.from_pretrained(ckpt, size={'longest_edge': R, 'shortest_edge': 0}, do_image_splitting=False)
    model = AutoModelForVision2Seq.from_pretrained(ckpt, torch_dtype=torch.float16).to(device); model.eval()
    text = processor.apply_chat_template(messages, add_generation_prompt=True)
    inputs = processor(text=text, images=images, return_tensors="pt")      -> input_ids, attention_mask, pixel_values, pixel_attention_mask
    ids = model.generate(**inputs, max_new_tokens=128);  processor.batch_decode(ids, skip_special_tokens=True)

``Idefics2Processor`` and ``Idefics2ForConditionalGeneration`` give exactly these calls.  The processor restates the
third-party Idefics2Processor / Idefics2ImageProcessor behaviour that matters on this path (absent from /root/reference:
transformers>=4.38.2, requirements.txt:16): the published chat template, the expansion of every ``<image>`` into
``<fake_token_around_image>`` + 64 x ``<image>`` + ``<fake_token_around_image>`` (adjacent images share the fake token), the
resize rule of ``size={'longest_edge': R, 'shortest_edge': 0}`` (bilinear, never enlarging — pinned by
tests/golden/idefics2_resize.json), rescale 1/255, normalise with mean = std = 0.5, zero padding to the largest image of the
sample with ``pixel_attention_mask``.  Tokenisation itself is the tokenizer's (third-party, not restated): pass any object with
``__call__(text, return_tensors=)``-style ``encode`` / ``batch_decode``; ``from_pretrained`` loads it with
``transformers.AutoTokenizer`` when the checkpoint has tokenizer files.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from .checkpoint import CheckpointSource
from .config import Idefics2Config, PerceiverConfig, TextConfig, VisionConfig
from .engine import KVCache
from .idefics2 import Idefics2Engine, Idefics2Weights, preprocess_image_u8
from .ops import Ops

IMAGE_TOKEN = "<image>"
FAKE_TOKEN = "<fake_token_around_image>"
END_OF_UTTERANCE = "<end_of_utterance>"


def apply_chat_template(messages: Sequence[dict], add_generation_prompt: bool = False) -> str:
    """The Idefics2 chat template (published with the checkpoints): ``Role:`` (+ space unless the turn starts with an image),
    the content items in order (text verbatim, every image as ``<image>``), ``<end_of_utterance>\\n`` after each turn,
    ``Assistant:`` when a generation prompt is asked for.  (IDEF:91)"""
    out = []
    for m in messages:
        content = m["content"]
        out.append(m["role"].capitalize())
        out.append(":" if content and content[0]["type"] == "image" else ": ")
        for item in content:
            if item["type"] == "text":
                out.append(item["text"])
            elif item["type"] == "image":
                out.append(IMAGE_TOKEN)
        out.append(END_OF_UTTERANCE + "\n")
    if add_generation_prompt:
        out.append("Assistant:")
    return "".join(out)


def expand_image_tokens(text: str, image_seq_len: int) -> str:
    """Every ``<image>`` becomes ``<fake><image> x L<fake>``; two fake tokens that meet between adjacent images collapse
    into one (third-party Idefics2Processor.__call__)."""
    block = FAKE_TOKEN + IMAGE_TOKEN * image_seq_len + FAKE_TOKEN
    return text.replace(IMAGE_TOKEN, block).replace(FAKE_TOKEN + FAKE_TOKEN, FAKE_TOKEN)


class Idefics2Processor:
    def __init__(self, tokenizer, longest_edge: int = 980, image_seq_len: int = 64):
        self.tokenizer, self.longest_edge, self.image_seq_len = tokenizer, longest_edge, image_seq_len

    @classmethod
    def from_pretrained(cls, path: str, size: Optional[dict] = None, do_image_splitting: bool = False, tokenizer=None, **unused):
        if do_image_splitting:
            raise NotImplementedError("do_image_splitting=True is not on the reference path (IDEF:25)")
        if size is not None and size.get("shortest_edge", 0) not in (0, None):
            raise NotImplementedError("only shortest_edge = 0, as in the reference script (IDEF:24)")
        edge = 980 if size is None else int(size["longest_edge"])
        if tokenizer is None:
            from transformers import AutoTokenizer          # third-party; needs the checkpoint's tokenizer files
            tokenizer = AutoTokenizer.from_pretrained(path)
        return cls(tokenizer, longest_edge=edge)

    def apply_chat_template(self, messages, add_generation_prompt: bool = False, **unused) -> str:
        return apply_chat_template(messages, add_generation_prompt)

    def image_arrays(self, images) -> List[np.ndarray]:
        """PIL images -> u8 HWC arrays at the processor's output size."""
        return [preprocess_image_u8(im, self.longest_edge) for im in images]

    def __call__(self, text: str, images=None, return_tensors: str = "pt", **unused) -> dict:
        if return_tensors != "pt":
            raise NotImplementedError("return_tensors='pt' only")
        images = list(images or [])
        if text.count(IMAGE_TOKEN) != len(images):
            raise ValueError(f"The number of images in the text {text.count(IMAGE_TOKEN)} and images {len(images)} should be the same.")
        enc = self.tokenizer(expand_image_tokens(text, self.image_seq_len), return_tensors="pt")
        out = {"input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"]}
        if images:
            arrs = self.image_arrays(images)
            H, W = max(a.shape[0] for a in arrs), max(a.shape[1] for a in arrs)
            pix = torch.zeros(1, len(arrs), 3, H, W, dtype=torch.float32)
            mask = torch.zeros(1, len(arrs), H, W, dtype=torch.int64)
            for i, a in enumerate(arrs):
                x = (torch.from_numpy(a.copy()).float() * (1.0 / 255.0) - 0.5) / 0.5        # rescale, then normalise (mean = std = 0.5)
                pix[0, i, :, :a.shape[0], :a.shape[1]] = x.permute(2, 0, 1)
                mask[0, i, :a.shape[0], :a.shape[1]] = 1
            out["pixel_values"], out["pixel_attention_mask"] = pix, mask
        return out

    def batch_decode(self, ids, skip_special_tokens: bool = True, **kw):
        return self.tokenizer.batch_decode(ids, skip_special_tokens=skip_special_tokens, **kw)


def load_idefics2_config(path: str) -> Idefics2Config:
    """``config.json`` of an Idefics2 checkpoint (HF layout: vision_config / text_config / perceiver_config)."""
    import dataclasses
    with open(os.path.join(path, "config.json")) as f:
        d = json.load(f)

    def pick(cls, src, rename=None):
        keys = {f.name for f in dataclasses.fields(cls)}
        src = dict(src or {})
        for a, b in (rename or {}).items():
            if a in src and b not in src:
                src[b] = src[a]
        return cls(**{k: v for k, v in src.items() if k in keys and v is not None and not isinstance(v, dict)})

    base = Idefics2Config()
    # overlay ONLY the keys the JSON carries onto the Idefics2 defaults (a diff-style saved config omits e.g. image_size = 980;
    # filling it from VisionConfig's own default, 364, would shrink the 70 x 70 position grid to 26 x 26)
    vkeys = {f.name for f in dataclasses.fields(VisionConfig)}
    vc = dataclasses.replace(base.vision_config, **{k: v for k, v in (d.get("vision_config") or {}).items()
                                                    if k in vkeys and v is not None and not isinstance(v, dict)})
    tc_src = dict(d.get("text_config") or {})
    tc = base.text_config if not tc_src else dataclasses.replace(base.text_config, **{
        k: v for k, v in tc_src.items() if k in {f.name for f in dataclasses.fields(TextConfig)} and k != "rope_scaling" and v is not None})
    pc = pick(PerceiverConfig, d.get("perceiver_config"), {"resampler_n_latents": "n_latents", "resampler_depth": "depth",
                                                           "resampler_n_heads": "n_heads", "resampler_head_dim": "head_dim"}) \
        if "perceiver_config" in d else base.perceiver_config
    return Idefics2Config(vision_config=vc, text_config=tc, perceiver_config=pc,
                          image_token_id=int(d.get("image_token_id", base.image_token_id)),
                          longest_edge=int(d.get("longest_edge", base.longest_edge)))


class Idefics2ForConditionalGeneration:
    """``AutoModelForVision2Seq.from_pretrained(...)`` stand-in for the reference script: ``.to()``, ``.eval()``, ``.device``,
    ``.generate(input_ids=, attention_mask=, pixel_values=, pixel_attention_mask=, max_new_tokens=)``."""

    def __init__(self, config: Idefics2Config, source_factory, compute_dtype=torch.float16, ops: Optional[Ops] = None,
                 eos_token_id: Sequence[int] = (2, 32002), bad_words_ids: Optional[Sequence[int]] = None):
        self.config, self._source_factory, self.compute_dtype, self._ops = config, source_factory, compute_dtype, ops
        self.eos_token_id = tuple(int(e) for e in eos_token_id)
        # stock Idefics2 generation_config suppresses <fake_token_around_image> and <image> (bad_words_ids): never generated
        self.bad_words_ids = tuple(int(b) for b in (bad_words_ids if bad_words_ids is not None
                                                    else (config.image_token_id - 1, config.image_token_id)))
        # see unpad_images.  "any" = the reference's own rule (`patches_subgrid.sum(...) > 0`: megatron_patch/model/idefics2/
        # idefics_vlm_model.py:608, language_model_llama3.py:654, and the transformers 4.4x releases Leopard-Idefics2 ran with);
        # "all" = transformers 5.x, kept for the golden fixture pinned to that version
        self.patch_validity = "any"
        self.precision = "fast"
        self._engine: Optional[Idefics2Engine] = None
        self.device = torch.device("cpu")

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=torch.float16, ops: Optional[Ops] = None, patch_validity: str = "any",
                        precision: Optional[str] = None, **unused):
        """``precision`` / ``LEOPARD_AMD_PRECISION``: see leopard_amd.compat.resolve_precision.  The reference loads this model in fp16
        (idefics2_multiimg.py:27-28), so the default for its request is the fast fp16 schedule; a float32 request selects lo4."""
        cfg = load_idefics2_config(path)
        eos, bad = (2, 32002), None
        gpath = os.path.join(path, "generation_config.json")
        if os.path.exists(gpath):
            with open(gpath) as f:
                g = json.load(f)
            e = g.get("eos_token_id", eos)
            eos = tuple(e) if isinstance(e, (list, tuple)) else (int(e),)
            if g.get("bad_words_ids") is not None:
                bad = [int(w[0]) for w in g["bad_words_ids"] if len(w) == 1]            # single-token bans (all stock Idefics2 has)
        dtype = torch_dtype if torch_dtype in (torch.float16, torch.bfloat16) else torch.float16
        m = cls(cfg, lambda dev, dt: CheckpointSource(path, dev, dt), dtype, ops, eos, bad)
        m.patch_validity = patch_validity
        from .compat import resolve_precision
        m.precision = resolve_precision(torch_dtype, dtype, precision)
        return m

    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if self._engine is None or device != self.device:
            ops = self._ops if self._ops is not None else Ops()
            W = Idefics2Weights.build(self.config, self._source_factory(device, self.compute_dtype), self.compute_dtype)
            self._engine = Idefics2Engine(self.config, W, ops=ops, device=device)
            if self.precision == "lo4" and not self._engine.lo4_supported():
                self.precision = "fast"                 # (the split mode is a LeopardEngine schedule; the Idefics2 tower has fast and lo4)
            if self.precision in ("fast", "lo4"):
                self._engine.precision = self.precision
            vocab = self.config.text_config.vocab_size
            bad = [b for b in self.bad_words_ids if 0 <= b < vocab]
            self._engine.suppress_tokens = torch.tensor(bad, dtype=torch.int64, device=device) if bad else None
            self.device = device
        return self

    @property
    def engine(self) -> Idefics2Engine:
        if self._engine is None:
            raise RuntimeError("call .to(device) first (the HIP engine is built on the target device)")
        return self._engine

    @staticmethod
    def unpad_images(pixel_values: torch.Tensor, pixel_attention_mask: Optional[torch.Tensor], patch_validity: str = "any",
                     patch: int = 14) -> List[torch.Tensor]:
        """[1, n, 3, H, W] (+ mask [1, n, H, W]) -> per-image fp32 [3, h', w']; an image that is entirely padding (all zeros) is
        dropped, as the third-party model does before its vision tower.

        ``patch_validity`` is the third-party rule for which patches of the padded canvas belong to an image (the tower then runs
        on exactly those patches, and the NaViT position ids are fractions of THEIR row / column counts):
          "all"  a patch counts when ALL its pixels are real (`patch mask sum == patch_size**2`: transformers 5.x, the version the
                 golden fixture tests/golden/idefics2_tiny.npz is pinned to) -> floor(h / P) x floor(w / P) patches, remainder
                 pixels dropped;
          "any"  (default) a patch counts when ANY of its pixels is real (`> 0`: the reference's own model code,
                 megatron_patch/model/idefics2/idefics_vlm_model.py:608, and the 4.4x releases Leopard-Idefics2 was run with,
                 requirements.txt:16) -> a partly zero-padded last patch row / column is kept wherever the common canvas has room
                 for it, i.e. for the smaller images of a mixed-size sample (fixture tests/golden/idefics2_tiny_any.npz).
        The crop returned here is (rows, cols) = patches x P in both cases, zero padding included under "any"."""
        if patch_validity not in ("all", "any"):
            raise ValueError("patch_validity must be 'all' or 'any'")
        imgs = []
        H, W = pixel_values.shape[-2], pixel_values.shape[-1]
        for i in range(pixel_values.shape[1]):
            x = pixel_values[0, i]
            if pixel_attention_mask is not None:
                m = pixel_attention_mask[0, i].to(torch.bool)
                if not bool(m.any()):
                    continue
                h, w = int(m.any(dim=1).sum()), int(m.any(dim=0).sum())
                if patch_validity == "any":
                    h, w = min(-(-h // patch) * patch, H // patch * patch), min(-(-w // patch) * patch, W // patch * patch)
                x = x[:, :h, :w]
            elif not bool((x != 0).any()):
                continue
            imgs.append(x.to(torch.float32).contiguous())
        return imgs

    @torch.no_grad()
    def generate(self, input_ids, attention_mask=None, pixel_values=None, pixel_attention_mask=None, max_new_tokens: int = 128,
                 eos_token_id=None, **unused) -> torch.Tensor:
        if input_ids.shape[0] != 1:
            raise NotImplementedError("batch 1 per call, as in the reference script (IDEF:88-97)")
        if attention_mask is not None and not bool(attention_mask.to(torch.bool).all()):
            raise NotImplementedError("padded prompts are not produced by the reference script (batch 1)")
        eng = self.engine
        images = None if pixel_values is None else self.unpad_images(pixel_values, pixel_attention_mask, self.patch_validity,
                                                                     self.config.vision_config.patch_size)
        eos = self.eos_token_id if eos_token_id is None else (tuple(eos_token_id) if isinstance(eos_token_id, (list, tuple)) else (int(eos_token_id),))
        ids = input_ids.reshape(1, -1)
        cache = eng._generation_cache(ids.shape[1] + max_new_tokens)           # one cache + captured decode graph per engine
        res = eng.prefill(ids.to(self.device), images, cache=cache)
        out = eng._greedy_loop([int(t) for t in ids.reshape(-1).tolist()], eng.first_token(res.logits_last), cache, max_new_tokens,
                               set(int(e) for e in eos))
        return torch.tensor([out], dtype=torch.long, device=input_ids.device)

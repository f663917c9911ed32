"""Deterministic synthetic parameters and inputs (no checkpoints, tokenizers or datasets exist offline).

Every value is ``odd_integer * 2**-e`` with |odd_integer| <= 255, i.e. exactly representable in bf16,
fp16 and fp32 alike, so "cast the weights once to the compute dtype" is lossless and the CPU oracle
(fp32) and the GPU path (bf16 / fp16) start from bit-identical numbers whatever the dtype.

The generator is counter based: element ``i`` of the tensor called ``name`` is a pure function of
``(crc32(name), i)`` — independent of construction order, torch version or device.  The same function
is implemented three times and cross-checked in tests: numpy here (CPU oracle side), and the HIP kernel
``lmi_fill_synthetic`` (leopard_amd/csrc/elementwise.hip) for the GPU box.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, Tuple

import numpy as np

from .config import LeopardConfig

# kinds of tensors (select the value law)
KIND_WEIGHT = 0     # (2b-255) * 2**-13                 std ~0.018  (linear / embedding / pos-emb / conv)
KIND_BIAS = 1       # (2b-255) * 2**-15                 std ~0.0045
KIND_NORM = 2       # (112 + (b>>3)) / 128              in [0.875, 1.117]  (LayerNorm / RMSNorm gain)


def name_seed(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def _mix32(x: np.ndarray) -> np.ndarray:
    """lowbias32 integer finaliser on uint32 (wrap-around arithmetic)."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def hash_bytes(seed: int, start: int, count: int) -> np.ndarray:
    """b(i) in [0, 255] for i = start .. start+count-1."""
    i = np.arange(start, start + count, dtype=np.uint64)
    lo = (i & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (i >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = _mix32(lo ^ np.uint32(seed))
        h = _mix32(h + hi * np.uint32(0x9E3779B9) + np.uint32(0x85EBCA6B))
    return (h >> np.uint32(24)).astype(np.int32)


def values_from_bytes(b: np.ndarray, kind: int) -> np.ndarray:
    b = b.astype(np.float32)
    if kind == KIND_WEIGHT:
        return (2.0 * b - 255.0) * np.float32(2.0 ** -13)
    if kind == KIND_BIAS:
        return (2.0 * b - 255.0) * np.float32(2.0 ** -15)
    if kind == KIND_NORM:
        return (112.0 + np.floor(b / 8.0)) * np.float32(1.0 / 128.0)
    raise ValueError(kind)


def synth_array(name: str, shape: Tuple[int, ...], kind: int) -> np.ndarray:
    """fp32 numpy tensor ``name`` (chunked so that 0.5 G-element tensors do not need 8x temporaries)."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    seed = name_seed(name)
    step = 1 << 24
    for s in range(0, n, step):
        c = min(step, n - s)
        out[s:s + c] = values_from_bytes(hash_bytes(seed, s, c), kind)
    return out.reshape(shape)


# --------------------------------------------------------------------------------------------------
# parameter inventory in the HF checkpoint key layout the reference's converter writes
# (Pai-Megatron-Patch/toolkits/model_checkpoints_convertor/llava/hf2megatron_llava.py:1050-1484:
#  language_model.model.*, language_model.lm_head.weight, multi_modal_projector.linear_{1,2}.*,
#  vision_tower.vision_model.*)
# --------------------------------------------------------------------------------------------------
def param_specs(cfg: LeopardConfig) -> Iterator[Tuple[str, Tuple[int, ...], int]]:
    vc, tc = cfg.vision_config, cfg.text_config
    v = "vision_tower.vision_model."
    yield v + "embeddings.patch_embedding.weight", (vc.hidden_size, vc.num_channels, vc.patch_size, vc.patch_size), KIND_WEIGHT
    yield v + "embeddings.patch_embedding.bias", (vc.hidden_size,), KIND_BIAS
    yield v + "embeddings.position_embedding.weight", (vc.num_patches, vc.hidden_size), KIND_WEIGHT
    for i in range(vc.num_hidden_layers):
        p = f"{v}encoder.layers.{i}."
        for ln in ("layer_norm1", "layer_norm2"):
            yield p + ln + ".weight", (vc.hidden_size,), KIND_NORM
            yield p + ln + ".bias", (vc.hidden_size,), KIND_BIAS
        for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield p + f"self_attn.{proj}.weight", (vc.hidden_size, vc.hidden_size), KIND_WEIGHT
            yield p + f"self_attn.{proj}.bias", (vc.hidden_size,), KIND_BIAS
        yield p + "mlp.fc1.weight", (vc.intermediate_size, vc.hidden_size), KIND_WEIGHT
        yield p + "mlp.fc1.bias", (vc.intermediate_size,), KIND_BIAS
        yield p + "mlp.fc2.weight", (vc.hidden_size, vc.intermediate_size), KIND_WEIGHT
        yield p + "mlp.fc2.bias", (vc.hidden_size,), KIND_BIAS
    yield v + "post_layernorm.weight", (vc.hidden_size,), KIND_NORM
    yield v + "post_layernorm.bias", (vc.hidden_size,), KIND_BIAS

    m = "multi_modal_projector."
    yield m + "linear_1.weight", (tc.hidden_size, cfg.projector_in), KIND_WEIGHT
    yield m + "linear_1.bias", (tc.hidden_size,), KIND_BIAS
    yield m + "linear_2.weight", (tc.hidden_size, tc.hidden_size), KIND_WEIGHT
    yield m + "linear_2.bias", (tc.hidden_size,), KIND_BIAS

    l = "language_model.model."
    hd = tc.head_dim
    yield l + "embed_tokens.weight", (tc.vocab_size, tc.hidden_size), KIND_WEIGHT
    for i in range(tc.num_hidden_layers):
        p = f"{l}layers.{i}."
        yield p + "input_layernorm.weight", (tc.hidden_size,), KIND_NORM
        yield p + "self_attn.q_proj.weight", (tc.num_attention_heads * hd, tc.hidden_size), KIND_WEIGHT
        yield p + "self_attn.k_proj.weight", (tc.num_key_value_heads * hd, tc.hidden_size), KIND_WEIGHT
        yield p + "self_attn.v_proj.weight", (tc.num_key_value_heads * hd, tc.hidden_size), KIND_WEIGHT
        yield p + "self_attn.o_proj.weight", (tc.hidden_size, tc.num_attention_heads * hd), KIND_WEIGHT
        yield p + "post_attention_layernorm.weight", (tc.hidden_size,), KIND_NORM
        yield p + "mlp.gate_proj.weight", (tc.intermediate_size, tc.hidden_size), KIND_WEIGHT
        yield p + "mlp.up_proj.weight", (tc.intermediate_size, tc.hidden_size), KIND_WEIGHT
        yield p + "mlp.down_proj.weight", (tc.hidden_size, tc.intermediate_size), KIND_WEIGHT
    yield l + "norm.weight", (tc.hidden_size,), KIND_NORM
    yield "language_model.lm_head.weight", (tc.vocab_size, tc.hidden_size), KIND_WEIGHT


def spec_table(cfg: LeopardConfig) -> Dict[str, Tuple[Tuple[int, ...], int]]:
    return {n: (s, k) for n, s, k in param_specs(cfg)}


def synth_state_dict_numpy(cfg: LeopardConfig) -> Dict[str, np.ndarray]:
    """All parameters as fp32 numpy (CPU oracle side; only sensible for reduced configs)."""
    return {n: synth_array(n, s, k) for n, s, k in param_specs(cfg)}


# --------------------------------------------------------------------------------------------------
# synthetic inputs (SURVEY.md 8d "Synthetic inputs")
# --------------------------------------------------------------------------------------------------
def synth_image_u8(seed: int, width: int, height: int) -> np.ndarray:
    """uint8 HWC noise image."""
    return np.random.default_rng(seed).integers(0, 256, (height, width, 3), dtype=np.uint8)


def synth_prompt_ids(vit_inputs_per_image, cfg: LeopardConfig, n_question: int = 32, seed: int = 0) -> np.ndarray:
    """Token ids with the structure the reference's prompt builder produces (EVAL:408-446): chat head,
    then per image a few text ids, an open marker, one image-token id per ViT input of that image and a
    close marker, then the question and the assistant tail.  Random ids never collide with the image
    token id."""
    rng = np.random.default_rng(1000 + seed)
    V = cfg.text_config.vocab_size
    hi = min(V, 128000) if V > 1000 else V
    img = cfg.image_token_index

    def rnd(n):
        x = rng.integers(3, hi, n)
        x[x == img] = 3
        return x.tolist()

    big = V >= 128256
    ids = [128000, 128006, 882, 128007, 271] if big else rnd(5)
    for k in vit_inputs_per_image:
        ids += rnd(4) + ([128025] if big else rnd(1)) + [img] * k + ([128026] if big else rnd(1))
    ids += rnd(n_question) + rnd(16)
    return np.asarray(ids, dtype=np.int64)


# --------------------------------------------------------------------------------------------------
# Leopard-Idefics2 parameter inventory in the HF key layout (Idefics2ForConditionalGeneration; the reference's
# converter: toolkits/model_checkpoints_convertor/idefics2/idefics2_hf2mg.py:1263-1662 — model.vision_model.*,
# model.connector.*, model.text_model.*, lm_head.weight)
# --------------------------------------------------------------------------------------------------
def idefics2_param_specs(cfg) -> Iterator[Tuple[str, Tuple[int, ...], int]]:
    vc, tc, pc = cfg.vision_config, cfg.text_config, cfg.perceiver_config
    v = "model.vision_model."
    yield v + "embeddings.patch_embedding.weight", (vc.hidden_size, vc.num_channels, vc.patch_size, vc.patch_size), KIND_WEIGHT
    yield v + "embeddings.patch_embedding.bias", (vc.hidden_size,), KIND_BIAS
    yield v + "embeddings.position_embedding.weight", (vc.num_patches, vc.hidden_size), KIND_WEIGHT
    for i in range(vc.num_hidden_layers):
        p = f"{v}encoder.layers.{i}."
        for ln in ("layer_norm1", "layer_norm2"):
            yield p + ln + ".weight", (vc.hidden_size,), KIND_NORM
            yield p + ln + ".bias", (vc.hidden_size,), KIND_BIAS
        for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield p + f"self_attn.{proj}.weight", (vc.hidden_size, vc.hidden_size), KIND_WEIGHT
            yield p + f"self_attn.{proj}.bias", (vc.hidden_size,), KIND_BIAS
        yield p + "mlp.fc1.weight", (vc.intermediate_size, vc.hidden_size), KIND_WEIGHT
        yield p + "mlp.fc1.bias", (vc.intermediate_size,), KIND_BIAS
        yield p + "mlp.fc2.weight", (vc.hidden_size, vc.intermediate_size), KIND_WEIGHT
        yield p + "mlp.fc2.bias", (vc.hidden_size,), KIND_BIAS
    yield v + "post_layernorm.weight", (vc.hidden_size,), KIND_NORM
    yield v + "post_layernorm.bias", (vc.hidden_size,), KIND_BIAS

    c = "model.connector."
    yield c + "modality_projection.gate_proj.weight", (tc.intermediate_size, vc.hidden_size), KIND_WEIGHT
    yield c + "modality_projection.up_proj.weight", (tc.intermediate_size, vc.hidden_size), KIND_WEIGHT
    yield c + "modality_projection.down_proj.weight", (tc.hidden_size, tc.intermediate_size), KIND_WEIGHT
    r = c + "perceiver_resampler."
    yield r + "latents", (pc.n_latents, tc.hidden_size), KIND_WEIGHT
    for i in range(pc.depth):
        p = f"{r}layers.{i}."
        for n in ("input_latents_norm", "input_context_norm", "post_attention_layernorm"):
            yield p + n + ".weight", (tc.hidden_size,), KIND_NORM
        yield p + "self_attn.q_proj.weight", (pc.n_heads * pc.head_dim, tc.hidden_size), KIND_WEIGHT
        yield p + "self_attn.k_proj.weight", (pc.num_key_value_heads * pc.head_dim, tc.hidden_size), KIND_WEIGHT
        yield p + "self_attn.v_proj.weight", (pc.num_key_value_heads * pc.head_dim, tc.hidden_size), KIND_WEIGHT
        yield p + "self_attn.o_proj.weight", (tc.hidden_size, pc.n_heads * pc.head_dim), KIND_WEIGHT
        yield p + "mlp.gate_proj.weight", (4 * tc.hidden_size, tc.hidden_size), KIND_WEIGHT
        yield p + "mlp.up_proj.weight", (4 * tc.hidden_size, tc.hidden_size), KIND_WEIGHT
        yield p + "mlp.down_proj.weight", (tc.hidden_size, 4 * tc.hidden_size), KIND_WEIGHT
    yield r + "norm.weight", (tc.hidden_size,), KIND_NORM

    l = "model.text_model."
    hd = tc.head_dim
    yield l + "embed_tokens.weight", (tc.vocab_size, tc.hidden_size), KIND_WEIGHT
    for i in range(tc.num_hidden_layers):
        p = f"{l}layers.{i}."
        yield p + "input_layernorm.weight", (tc.hidden_size,), KIND_NORM
        yield p + "self_attn.q_proj.weight", (tc.num_attention_heads * hd, tc.hidden_size), KIND_WEIGHT
        yield p + "self_attn.k_proj.weight", (tc.num_key_value_heads * hd, tc.hidden_size), KIND_WEIGHT
        yield p + "self_attn.v_proj.weight", (tc.num_key_value_heads * hd, tc.hidden_size), KIND_WEIGHT
        yield p + "self_attn.o_proj.weight", (tc.hidden_size, tc.num_attention_heads * hd), KIND_WEIGHT
        yield p + "post_attention_layernorm.weight", (tc.hidden_size,), KIND_NORM
        yield p + "mlp.gate_proj.weight", (tc.intermediate_size, tc.hidden_size), KIND_WEIGHT
        yield p + "mlp.up_proj.weight", (tc.intermediate_size, tc.hidden_size), KIND_WEIGHT
        yield p + "mlp.down_proj.weight", (tc.hidden_size, tc.intermediate_size), KIND_WEIGHT
    yield l + "norm.weight", (tc.hidden_size,), KIND_NORM
    yield "lm_head.weight", (tc.vocab_size, tc.hidden_size), KIND_WEIGHT


def idefics2_state_dict_numpy(cfg) -> Dict[str, np.ndarray]:
    return {n: synth_array(n, s, k) for n, s, k in idefics2_param_specs(cfg)}

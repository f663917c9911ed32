"""Parameter sources and the device-side weight layout of the engine.

``SynthSource`` materialises the deterministic synthetic parameters of leopard_amd.synth directly on the GPU
with ``lmi_fill_synthetic`` (bit-identical to the numpy generator the CPU oracle uses).  ``TensorSource`` wraps
an in-memory state dict (tests, HF checkpoints loaded by leopard_amd.checkpoint).  Both speak the HF key layout
written by the reference's converter (toolkits/model_checkpoints_convertor/llava/hf2megatron_llava.py:1050-1484).

``EngineWeights`` holds what the kernels consume: 16-bit GEMM operands fused / padded / interleaved once at load
time (q|k|v concatenated; gate/up interleaved in 32-row blocks for the SwiGLU epilogue; N padded to 128 and K
to 64 with zeros), fp32 biases / norm gains / position embedding.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import torch

from .config import LeopardConfig
from .synth import KIND_WEIGHT, name_seed, spec_table


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class SynthSource:
    def __init__(self, cfg: LeopardConfig, ops, device, dtype):
        self.cfg, self.ops, self.device, self.dtype = cfg, ops, device, dtype
        self.specs = spec_table(cfg)

    def get(self, name: str) -> torch.Tensor:
        shape, kind = self.specs[name]
        dt = self.dtype if kind == KIND_WEIGHT else torch.float32
        out = torch.empty(shape, dtype=dt, device=self.device)
        self.ops.fill_synthetic(out, name_seed(name), kind)
        return out


class TensorSource:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device, dtype):
        self.sd, self.device, self.dtype = state_dict, device, dtype

    def get(self, name: str) -> torch.Tensor:
        t = self.sd[name]
        if not torch.is_tensor(t):
            t = torch.from_numpy(t)
        is_matrix = t.dim() >= 2
        return t.to(device=self.device, dtype=self.dtype if is_matrix else torch.float32)


def _pad2(w: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    if w.shape == (rows, cols):
        return w.contiguous()
    out = torch.zeros(rows, cols, dtype=w.dtype, device=w.device)
    out[:w.shape[0], :w.shape[1]] = w
    return out


def _pad1(b: torch.Tensor, n: int) -> torch.Tensor:
    b = b.to(torch.float32)
    if b.numel() == n:
        return b.contiguous()
    out = torch.zeros(n, dtype=torch.float32, device=b.device)
    out[:b.numel()] = b
    return out


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[F,K],[F,K] -> [2F,K] with rows in blocks of 32: g[0:32] u[0:32] g[32:64] u[32:64] ... (F % 32 == 0)."""
    F, K = gate.shape
    return torch.stack([gate.view(F // 32, 32, K), up.view(F // 32, 32, K)], dim=1).reshape(2 * F, K).contiguous()


def patch_weight_image_order(w: torch.Tensor, patch: int) -> torch.Tensor:
    """Conv weight [D, 3, P, P] -> [D, KP] in the K order lmi_patch_embed stages pixels in: k = ky * RP + kx * 3 + c (the byte order
    of an HWC image row), every pixel row padded from 3P to RP = roundup(3P, 8) and the whole row to a multiple of 64, zeros in
    the pad positions."""
    D = w.shape[0]
    rp = (3 * patch + 7) & ~7
    kp = _round_up(patch * rp, 64)
    rows = w.reshape(D, 3, patch, patch).permute(0, 2, 3, 1).reshape(D, patch, 3 * patch)        # [D, ky, (kx, c)]
    out = torch.zeros(D, kp, dtype=w.dtype, device=w.device)
    out[:, :patch * rp].view(D, patch, rp)[:, :, :3 * patch] = rows
    return out


def rope_permute_rows(w: torch.Tensor, head_dim: int = 128) -> torch.Tensor:
    """Row order of the q / k projection weights for lmi_rmsnorm_rope: inside every head the rows d = 0..127 are stored as
    [0..31, 64..95, 32..63, 96..127], so that each 64-column slice of the GEMM output that one wave owns holds 32 first-half
    elements next to their rotate-half partners (d, d + 64).  The kernel's epilogue restores the natural order on store."""
    assert head_dim == 128 and w.shape[0] % head_dim == 0
    idx = torch.cat([torch.arange(0, 32), torch.arange(64, 96), torch.arange(32, 64), torch.arange(96, 128)]).to(w.device)
    n = w.shape[0] // head_dim
    return w.view(n, head_dim, -1)[:, idx].reshape(w.shape).contiguous()


def skinny_pack(w: torch.Tensor) -> torch.Tensor:
    """[N, K] weight -> the same elements in lmi_gemm_skinny's packed order (N % 16 == 0, K % 128 == 0): 1-KiB blocks
    [16-row group r][k-step s of 128][32-k chunk c][lane l = 16 g + i][8 elements] with element (r, s, c, g, i, j) = w[16 r + i, 128 s + 32 c + 8 g + j]
    — exactly what lane l of a wave feeds v_mfma_f32_16x16x32 as its A operand, so a wave-wide load is one contiguous 1-KiB request."""
    N, K = w.shape
    assert N % 16 == 0 and K % 128 == 0
    v = w.contiguous().view(N // 16, 16, K // 128, 4, 4, 8)            # (r, i, s, c, g, j)
    return v.permute(0, 2, 3, 4, 1, 5).contiguous().view(N, K)          # (r, s, c, g, i, j)


def skinny_unpack(w: torch.Tensor) -> torch.Tensor:
    """Inverse of skinny_pack: the row-major [N, K] matrix again."""
    N, K = w.shape
    v = w.contiguous().view(N // 16, K // 128, 4, 4, 16, 8)            # (r, s, c, g, i, j)
    return v.permute(0, 4, 1, 2, 3, 5).contiguous().view(N, K)          # (r, i, s, c, g, j)


# One copy of the LLM weights for every kernel that reads them: the packed order is a permutation of the 16-byte pieces of the row-major
# matrix that keeps 16-row groups and 64-k tiles together, so the prefill GEMM's LDS-DMA stages the SAME LDS image from it (ldw =
# LMI_LDW_PACKED(K), csrc/gemm.h GemmStager) and the decode kernels (lmi_gemm_skinny, packed = 1) stream it in coalesced 1-KiB requests.
# A packed tensor keeps its [N, K] shape and carries a mark that leopard_amd.ops reads when it builds the call.
def mark_packed(w: torch.Tensor) -> torch.Tensor:
    w._lmi_packed = True
    return w


def is_packed(w) -> bool:
    return bool(getattr(w, "_lmi_packed", False))


def packable(w: torch.Tensor) -> bool:
    return w.dim() == 2 and w.shape[0] % 16 == 0 and w.shape[1] % 128 == 0 and w.element_size() == 2


def as_packed(w: torch.Tensor) -> torch.Tensor:
    return w if is_packed(w) else mark_packed(skinny_pack(w))


def as_row_major(w: torch.Tensor) -> torch.Tensor:
    """The nn.Linear layout of a weight whichever way it is stored (a fresh tensor when it was packed)."""
    return skinny_unpack(w) if is_packed(w) else w


@dataclass
class VitLayerW:
    ln1_w: torch.Tensor; ln1_b: torch.Tensor
    qkv_w: torch.Tensor; qkv_b: torch.Tensor
    o_w: torch.Tensor; o_b: torch.Tensor
    ln2_w: torch.Tensor; ln2_b: torch.Tensor
    fc1_w: torch.Tensor; fc1_b: torch.Tensor
    fc2_w: torch.Tensor; fc2_b: torch.Tensor


@dataclass
class LlmLayerW:
    in_norm: torch.Tensor
    qkv_w: torch.Tensor
    o_w: torch.Tensor
    post_norm: torch.Tensor
    gu_w: torch.Tensor
    down_w: torch.Tensor
    qkv_w_rope: torch.Tensor = None        # qkv_w with the q / k rows in lmi_rmsnorm_rope's order (head_dim 128 only)



def build_llm_layers(g, prefix: str, tc, tp_rank: int = 0, tp_size: int = 1) -> List[LlmLayerW]:
    """The Llama / Mistral decoder layers as the kernels consume them (shared by Leopard-LLaVA and Leopard-Idefics2): q|k|v fused
    (+ the copy in lmi_rmsnorm_rope's row order), gate/up interleaved, and — for ``tp_size`` > 1 — the Megatron tensor-parallel
    shard ``tp_rank``: q/k/v and gate/up split by output rows (whole heads / FFN slices), o_proj and down_proj by input columns."""
    Dt, full_q, full_kv = tc.hidden_size, tc.num_attention_heads * tc.head_dim, tc.num_key_value_heads * tc.head_dim
    heads, kv_heads, ff = tc.num_attention_heads // tp_size, tc.num_key_value_heads // tp_size, tc.intermediate_size // tp_size

    def expect(name: str, shape) -> torch.Tensor:
        """A checkpoint whose config.json disagrees with its tensors (the reference converter writes
        num_key_value_heads = num_attention_heads for non-70b models, hf2megatron_llava.py:1035) must fail here, not
        slice silently and leave attention reading unwritten K/V columns."""
        t = g(name)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(t.shape)} does not match the configuration (expected {tuple(shape)}: "
                             f"hidden {Dt}, {tc.num_attention_heads} q / {tc.num_key_value_heads} kv heads x {tc.head_dim}, "
                             f"FFN {tc.intermediate_size})")
        return t
    layers = []
    for i in range(tc.num_hidden_layers):
        p = f"{prefix}layers.{i}."
        qw, kw = heads * tc.head_dim, kv_heads * tc.head_dim
        rq, rk, rf = slice(tp_rank * qw, (tp_rank + 1) * qw), slice(tp_rank * kw, (tp_rank + 1) * kw), slice(tp_rank * ff, (tp_rank + 1) * ff)
        qkv_w = torch.cat([expect(p + "self_attn.q_proj.weight", (full_q, Dt))[rq], expect(p + "self_attn.k_proj.weight", (full_kv, Dt))[rk],
                           expect(p + "self_attn.v_proj.weight", (full_kv, Dt))[rk]], dim=0).contiguous()
        qkv_rope = None
        if tc.head_dim == 128:
            qkv_rope = torch.cat([rope_permute_rows(qkv_w[:qw + kw]), qkv_w[qw + kw:]], dim=0).contiguous()
        layers.append(LlmLayerW(
            in_norm=g(p + "input_layernorm.weight").float().contiguous(),
            qkv_w=qkv_w, qkv_w_rope=qkv_rope,
            o_w=expect(p + "self_attn.o_proj.weight", (Dt, full_q))[:, rq].contiguous(),
            post_norm=g(p + "post_attention_layernorm.weight").float().contiguous(),
            gu_w=interleave_gate_up(expect(p + "mlp.gate_proj.weight", (tc.intermediate_size, Dt))[rf],
                                    expect(p + "mlp.up_proj.weight", (tc.intermediate_size, Dt))[rf]),
            down_w=expect(p + "mlp.down_proj.weight", (Dt, tc.intermediate_size))[:, rf].contiguous()))
    return layers


def check_tp_degree(tc, tp_size: int) -> None:
    if tp_size > 1 and (tc.num_attention_heads % tp_size or tc.num_key_value_heads % tp_size or (tc.intermediate_size // tp_size) % 64
                        or tc.intermediate_size % tp_size):
        raise ValueError(f"tensor parallel degree {tp_size} must divide the head counts ({tc.num_attention_heads}/"
                         f"{tc.num_key_value_heads}) and leave an FFN slice that is a multiple of 64")

@dataclass
class EngineWeights:
    cfg: LeopardConfig
    dtype: torch.dtype
    patch_w_fused: torch.Tensor = None     # the conv weight in lmi_patch_embed's image K order (patch_weight_image_order)
    patch_b: torch.Tensor = None; pos_emb: torch.Tensor = None
    vit_layers: List[VitLayerW] = field(default_factory=list)
    post_ln_w: torch.Tensor = None; post_ln_b: torch.Tensor = None
    proj1_w: torch.Tensor = None; proj1_b: torch.Tensor = None
    proj2_w: torch.Tensor = None; proj2_b: torch.Tensor = None
    embed: torch.Tensor = None
    llm_layers: List[LlmLayerW] = field(default_factory=list)
    final_norm: torch.Tensor = None
    lm_head: torch.Tensor = None
    # padded geometry
    vit_ff: int = 0         # fc1 width padded to 128
    llm_ff: int = 0
    # tensor-parallel shard of the LLM (1 = whole model): local head counts, llm_ff is the local FFN slice
    llm_heads: int = 0
    llm_kv_heads: int = 0
    tp_rank: int = 0
    tp_size: int = 1

    @classmethod
    def build(cls, cfg: LeopardConfig, source, dtype, tp_rank: int = 0, tp_size: int = 1) -> "EngineWeights":
        """``tp_size`` > 1: Megatron-style tensor-parallel shard ``tp_rank`` of the LLM (SURVEY.md 8e phase B): q/k/v and gate/up
        split by output rows (whole heads / FFN slices), o_proj and down_proj by input columns; norms, embedding, lm_head and
        the whole vision side are replicated.  The engine all-reduces the two partial products per layer (leopard_amd.dist)."""
        vc, tc = cfg.vision_config, cfg.text_config
        W = cls(cfg=cfg, dtype=dtype)
        W.tp_rank, W.tp_size = tp_rank, tp_size
        check_tp_degree(tc, tp_size)
        for dim, what in ((vc.hidden_size, "vision hidden"), (tc.hidden_size, "text hidden"),
                          (tc.num_attention_heads * tc.head_dim, "q width"), (tc.num_key_value_heads * tc.head_dim, "kv width")):
            if dim % 128:
                raise ValueError(f"{what} size {dim} must be a multiple of 128 for the MFMA GEMM tiles")
        if tc.intermediate_size % 64:
            raise ValueError("LLM FFN width must be a multiple of 64")
        g = source.get
        v = "vision_tower.vision_model."
        W.vit_ff = _round_up(vc.intermediate_size, 128)
        W.llm_ff = tc.intermediate_size // tp_size
        W.llm_heads, W.llm_kv_heads = tc.num_attention_heads // tp_size, tc.num_key_value_heads // tp_size
        W.patch_w_fused = patch_weight_image_order(g(v + "embeddings.patch_embedding.weight"), vc.patch_size)
        W.patch_b = _pad1(g(v + "embeddings.patch_embedding.bias"), vc.hidden_size)
        W.pos_emb = g(v + "embeddings.position_embedding.weight").to(torch.float32).contiguous()
        for i in range(vc.num_hidden_layers):
            p = f"{v}encoder.layers.{i}."
            qkv_w = torch.cat([g(p + f"self_attn.{n}_proj.weight") for n in "qkv"], dim=0).contiguous()
            qkv_b = torch.cat([g(p + f"self_attn.{n}_proj.bias").to(torch.float32) for n in "qkv"], dim=0).contiguous()
            W.vit_layers.append(VitLayerW(
                ln1_w=g(p + "layer_norm1.weight").float().contiguous(), ln1_b=g(p + "layer_norm1.bias").float().contiguous(),
                qkv_w=_pad2(qkv_w, _round_up(qkv_w.shape[0], 128), vc.hidden_size), qkv_b=_pad1(qkv_b, _round_up(qkv_b.numel(), 128)),
                o_w=g(p + "self_attn.out_proj.weight").contiguous(), o_b=_pad1(g(p + "self_attn.out_proj.bias"), vc.hidden_size),
                ln2_w=g(p + "layer_norm2.weight").float().contiguous(), ln2_b=g(p + "layer_norm2.bias").float().contiguous(),
                fc1_w=_pad2(g(p + "mlp.fc1.weight"), W.vit_ff, vc.hidden_size), fc1_b=_pad1(g(p + "mlp.fc1.bias"), W.vit_ff),
                fc2_w=_pad2(g(p + "mlp.fc2.weight"), vc.hidden_size, W.vit_ff), fc2_b=_pad1(g(p + "mlp.fc2.bias"), vc.hidden_size)))
        W.post_ln_w = g(v + "post_layernorm.weight").float().contiguous()
        W.post_ln_b = g(v + "post_layernorm.bias").float().contiguous()
        m = "multi_modal_projector."
        W.proj1_w = g(m + "linear_1.weight").contiguous(); W.proj1_b = _pad1(g(m + "linear_1.bias"), tc.hidden_size)
        W.proj2_w = g(m + "linear_2.weight").contiguous(); W.proj2_b = _pad1(g(m + "linear_2.bias"), tc.hidden_size)
        l = "language_model.model."
        W.embed = g(l + "embed_tokens.weight").contiguous()
        W.llm_layers = build_llm_layers(g, l, tc, tp_rank, tp_size)
        W.final_norm = g(l + "norm.weight").float().contiguous()
        head = g("language_model.lm_head.weight")
        W.lm_head = _pad2(head, _round_up(head.shape[0], 128), tc.hidden_size)
        return W

    def nbytes(self) -> int:
        n = 0
        def add(t):
            nonlocal n
            if torch.is_tensor(t):
                n += t.numel() * t.element_size()
        for k, val in self.__dict__.items():
            if isinstance(val, list):
                for lay in val:
                    for t in lay.__dict__.values():
                        add(t)
            else:
                add(val)
        return n

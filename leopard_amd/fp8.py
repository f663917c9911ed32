"""fp8 (OCP e4m3fn) schedule for the linears of the SigLIP and Llama layers — BASELINE.json configs[4] ("fp8 linears"), SURVEY.md 8
row a5/a11 at reduced operand precision.  NEVER the headline: bench.py reports it as its own dtype.

What is fp8: the A operand and the weight of the four linears of every ViT layer (qkv, out_proj, fc1, fc2) and every LLM layer
(qkv, o_proj, gate/up, down_proj), multiplied by v_mfma_scale_f32_32x32x64_f8f6f4 with fp32 accumulation (lmi_gemm_fp8).
What is not: the residual stream (fp32), q / k / v, the attention (16-bit), the patch embed, the projector, the head, the decode.
Hand-overs are fused into the producers: LayerNorm / RMSNorm write fp8 (lmi_norm_fp8), GELU / SwiGLU epilogues write fp8, the attention
kernel writes its output as the o_proj operand (lmi_attn_varlen_fwd_fp8), and the Llama q|k|v GEMM rotates and appends to the KV cache
in its epilogue (lmi_rope_qkv_fp8) — no conversion or RoPE launches are left (engine.fp8_fused = False restores them for A/B).

Scales are static powers of two (an E8M0 exponent the MFMA applies for free):
  * weights: per tensor, amax mapped into (224, 448];
  * activations: per site (layer x {norm1 out, attention out, norm2 out, MLP act out}), from the amax a calibration prefill of the
    16-bit path saw, with 2x headroom; conversion saturates (lmi_norm_fp8, lmi_quantize_fp8, the fp8-output GEMM epilogues).
e4m3 has the same relative precision over its whole normal range, so the exponent choice only has to avoid saturation."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch

F8_MAX = 448.0
VIT_SITES = ("h1", "att", "h2", "ff")
LLM_SITES = ("h1", "att", "h2", "gu")
ATTN_SITES = ("q", "k", "v")          # operands of the fp8 attention arithmetic (engine.fp8_attention): rotated q / k and v of every Llama layer


def pow2_exp(amax: float, headroom: float) -> int:
    """e with amax * headroom * 2^e in (224, 448]."""
    if not amax > 0.0 or not math.isfinite(amax):
        return 0
    return int(math.floor(math.log2(F8_MAX / (headroom * amax))))


@dataclass
class Fp8Linear:
    w8: torch.Tensor           # uint8 [N, K]: fp8(w * 2^e)
    e: int


@dataclass
class Fp8Layer:
    lin: Dict[str, Fp8Linear] = field(default_factory=dict)
    act: Dict[str, int] = field(default_factory=dict)          # site -> exponent: the operand is fp8(value * 2^act)

    def out_exp(self, site: str, name: str) -> int:
        """Exponent the GEMM applies to its accumulators to undo both operand scales."""
        return -(self.act[site] + self.lin[name].e)


@dataclass
class Fp8Plan:
    vit: List[Fp8Layer] = field(default_factory=list)
    llm: List[Fp8Layer] = field(default_factory=list)

    def nbytes(self) -> int:
        return sum(l.w8.numel() for lay in self.vit + self.llm for l in lay.lin.values())


def quantize_linear(ops, w: torch.Tensor) -> Fp8Linear:
    if w.shape[1] % 128 or w.shape[0] % 128:
        raise ValueError(f"fp8 linears need N % 128 == 0 and K % 128 == 0, got weight {tuple(w.shape)}")
    e = pow2_exp(float(w.abs().max()), 1.0)
    w8 = torch.empty(w.shape, dtype=torch.uint8, device=w.device)
    ops.quantize_fp8(w, w8, 2.0 ** e)
    return Fp8Linear(w8, e)


def calibrate(engine, samples, headroom: float = 2.0) -> Fp8Plan:
    """``samples``: [(input_ids, tiles)] run through the 16-bit path with amax recorders at the operand sites."""
    if engine.tp_size != 1:
        raise NotImplementedError("the fp8 schedule is single-rank (replica mode)")
    amax: Dict[tuple, float] = {}

    def rec(site, t):
        amax[site] = max(amax.get(site, 0.0), float(t.abs().max()))
    old = (engine._rec, engine.fuse_norm_rope, engine.fp8, engine.graph_encode)
    engine._rec, engine.fuse_norm_rope, engine.fp8, engine.graph_encode = rec, False, None, False
    try:
        for ids, tiles in samples:
            engine.prefill(ids, tiles)
    finally:
        engine._rec, engine.fuse_norm_rope, engine.fp8, engine.graph_encode = old
    W, ops, plan = engine.W, engine.ops, Fp8Plan()
    for li, L in enumerate(W.vit_layers):
        from .weights import as_row_major as _rm                 # engine.pack_vit_weights may have packed them
        lay = Fp8Layer(lin={"qkv": quantize_linear(ops, _rm(L.qkv_w)), "o": quantize_linear(ops, _rm(L.o_w)),
                            "fc1": quantize_linear(ops, _rm(L.fc1_w)), "fc2": quantize_linear(ops, _rm(L.fc2_w))})
        for s in VIT_SITES:
            lay.act[s] = pow2_exp(amax.get(("vit", li, s), 0.0), headroom)
        plan.vit.append(lay)
    from .weights import as_row_major                       # the 16-bit weights may be stored in the packed order (engine.pack_llm_weights)
    for li, L in enumerate(W.llm_layers):
        lay = Fp8Layer(lin={"qkv": quantize_linear(ops, engine._qkv_natural(L)), "o": quantize_linear(ops, as_row_major(L.o_w)),
                            "gu": quantize_linear(ops, as_row_major(L.gu_w)), "down": quantize_linear(ops, as_row_major(L.down_w))})
        if getattr(L, "qkv_w_rope", None) is not None:      # the same rows in rope_permute_rows order: q|k|v + RoPE + KV append in ONE fp8 launch
            lay.lin["qkv_rope"] = quantize_linear(ops, as_row_major(L.qkv_w_rope))
            assert lay.lin["qkv_rope"].e == lay.lin["qkv"].e
        for s in LLM_SITES:
            lay.act[s] = pow2_exp(amax.get(("llm", li, s), 0.0), headroom)
        for s in ATTN_SITES:
            if ("llm", li, s) in amax:
                lay.act[s] = pow2_exp(amax[("llm", li, s)], headroom)
        plan.llm.append(lay)
    return plan

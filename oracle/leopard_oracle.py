"""CPU ORACLE for the Leopard-LLaVA multi-image prefill path  —  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the CPU timing baseline.  Nothing under ``leopard_amd/`` imports it;
the product path fails loudly when the HIP extension is missing.

What this is: a plain PyTorch fp32 restatement (no ``transformers`` import, no reference files) of
every arithmetic step on the reference's inference hot path.  Abbreviations in the citations:
  EVAL = /root/reference/evaluations/models/llava_multiimg_siglip_anyres.py
  ROPE = /root/reference/Pai-Megatron-Patch/Megatron-LM-240603/megatron/core/models/common/embeddings/rotary_pos_embedding.py
  RMS  = /root/reference/Pai-Megatron-Patch/Megatron-LM-240603/megatron/legacy/model/rms_norm.py
  XFMR = /root/reference/Pai-Megatron-Patch/megatron_patch/model/llava/transformer.py
  VLM  = /root/reference/Pai-Megatron-Patch/megatron_patch/model/llava/vlm_model.py
  IVT  = /root/reference/Pai-Megatron-Patch/megatron_patch/model/idefics2/idefics_vision_tower.py

Where the arithmetic lives: the Leopard-authored pieces (tiler, pixel_shuffle, projector, forward glue)
are in EVAL; the ViT, the LLM and the image/text merge live in the un-vendored third-party dependency
``transformers>=4.38.2`` (requirements.txt:16) — SiglipVisionModel, LlamaForCausalLM and
LlavaForConditionalGeneration._merge_input_ids_with_image_features (4.38 .. 4.4x).  Their published
algorithms are restated here and cross-read with the in-tree Megatron analogues cited per function.

Pinning: the reference ships NO tests or golden vectors for this path.  This oracle is
pinned instead against outputs of the reference itself run in the build container — the reference's
own ``allocate_patches / select_best_resolution / resize_and_pad_image / divide_to_patches /
pixel_shuffle / myLlavaMultiModalProjector`` imported directly, and the reference
``myLlavaForConditionalGeneration.forward`` executed UNMODIFIED (unbound, over a shim holding
third-party transformers-5.15 SigLIP / Llama modules) — captured as fixtures under ``tests/golden/`` by
``oracle/gen_golden.py`` (committed).  ``tests/test_oracle_golden.py`` checks this file against them:
integers / index maps bit-exact, fp32 tensors <= 1e-5.  One caveat stays open: the 4.38 merge routine
is not available offline, so the merge fixture pins this restatement against the survey's restatement
run through the reference forward, not against a 4.38 install ("pinned by construction").
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ==================================================================================================
# Rounding-point emulation and per-layer tracing (both OFF by default: the oracle is plain fp32).
#
# The HIP path feeds 16-bit operands to the MFMAs (fp32 accumulate, fp32 residual stream), so it
# differs from this fp32 restatement by one rounding at each point where an activation is handed from
# one kernel to the next.  ``emulate_rounding(dtype)`` rounds this oracle's activations to ``dtype``
# (fp16 / bf16) at exactly those hand-over points and nowhere else.  Tests use it two ways:
#   * oracle(emulated) vs oracle(fp32)  = the PREDICTED error budget of 16-bit operands at a given depth;
#   * HIP vs oracle(emulated)           = what is left (accumulation order, exp/rsqrt approximations):
#                                         an order of magnitude smaller when the kernels are right.
# ``trace`` (a list) receives (name, tensor) after the embeddings and after every layer.
# ==================================================================================================
class _Emu:
    dtype = None          # None = no emulation
    trace = None          # list or None
    fp32_head = True      # the HIP path feeds the last-token lm_head the fp32 normalised row (lmi_lm_head_last)
    fused = True          # production Llama schedule (lmi_gemm_ex / lmi_rmsnorm_rope): the RMSNorm operand is rounded as
                          # T(x * gamma) BEFORE the row scale, and q / k are rounded once, after the rotation
    exact_sites = frozenset()   # split-operand study (tools/split_operand_study.py): hand-over sites treated as EXACT, i.e. as if the
                          # operand were handed over as a hi + lo pair of 16-bit values: "norm" (LayerNorm / RMSNorm outputs feeding
                          # q|k|v, fc1, gate/up), "attn_out" (o_proj / out_proj operand), "mlp_act" (GELU / SwiGLU output feeding fc2 / down_proj)
    fp8_block = 0         # 0 = per-tensor fp8 scales (what the engine does); 32 = per-32-element E8M0 block scales (study only)
    operand_dtype = None  # fp8 schedule (BASELINE config 5): the A operands AND the weights of the ViT / LLM layer linears are
                          # float8_e4m3fn with per-tensor power-of-two scales; everything else stays at ``dtype``
    fp8_attention = False # fp8 schedule with engine.fp8_attention (csrc/attention_fp8.h): the rotated q / k, v (per-tensor power-of-two
                          # scales) and the probabilities entering P.V of the LLM layers are e4m3 as well; row sums and accumulators fp32
    lo_sites = frozenset()  # low-bit correction pass (round 5; csrc/gemm.h second k-loop phase, engine.precision = "lo4"): at these
                          # hand-over sites the A operand is handed over as T(x) PLUS an MX block-scaled low-bit image of the rounding
                          # residual x - T(x) (per-32-element E8M0 scale along the contraction axis), multiplied with a low-bit image of
                          # the weight (one E8M0 scale per weight row) into the same fp32 accumulators
    tower = ""            # "vit" / "llm" while that stack runs: lo_sites / exact_sites entries may be tower-qualified ("llm.mlp_act")
    layer = -1            # index of the running layer: lo_sites entries may carry a layer range ("llm.mlp_act@0-15"; round 6, the
                          # per-site correction policy of engine.lo4_policy / tools/lo4_policy_study.py)
    sub = ""              # which of a layer's two norm operands is being handed over: "norm1" (q|k|v) / "norm2" (fc1, gate/up) — an
                          # entry "norm" selects both
    lo_row_start = 0      # round 6 (engine.lo4_rows): the correction is applied to the rows >= lo_row_start of the LLM stream only — the rows
                          # whose logits are read; the other rows' roundings reach them through the softmax average over ~S keys
    lo_fmt = "e2m1"       # element format of the residual and of the weight image: "e2m1" (fp4), "e2m3" / "e3m2" (fp6), "e4m3" (fp8)
    lo_wblock = 0         # 0 = one scale per weight row (what the engine does); 32 = per-32 block scales on the weight image (study)
    _wcache: dict = {}


class _Split:
    """A operand handed over as a 16-bit value plus a low-bit image of its rounding residual (see _Emu.lo_sites)."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo):
        self.hi, self.lo = hi, lo

    def scaled(self, f):                 # per-row factor applied to the accumulator rows (folded RMSNorm)
        return _Split(self.hi * f, self.lo * f)


_LO_GRID = {"e2m1": (1, 2, 6.0), "e2m3": (3, 2, 7.5), "e3m2": (2, 4, 28.0), "e4m3": (3, 8, 448.0)}   # mantissa bits, emax, max normal


def _lo_round(x: Tensor, fmt: str, block: int) -> Tensor:
    """MX-style quantisation along the last axis: shared scale 2^(floor(log2(amax)) - emax) per ``block`` elements (block = 0: per row),
    elements rounded to nearest even on the format's grid (subnormals included) and saturated at its largest normal."""
    mbits, emax, vmax = _LO_GRID[fmt]
    if block and x.shape[-1] % block:                # the kernels' images are zero-padded to whole blocks (SigLIP's 4304-wide fc1 output)
        pad = block - x.shape[-1] % block
        return _lo_round(F.pad(x, (0, pad)), fmt, block)[..., :x.shape[-1]]
    shp = x.shape
    b = x.reshape(*shp[:-1], shp[-1] // block, block) if block else x.unsqueeze(-2)
    amax = b.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(2.0 ** -120))) - emax
    sc = torch.exp2(e)
    y = (b / sc).clamp(-vmax, vmax)
    # grid step of the binade of |y| (minimum normal exponent 0 for e2m1 / e2m3, i.e. subnormal step 2^-mbits)
    emin = {"e2m1": 0, "e2m3": 0, "e3m2": -2, "e4m3": -6}[fmt]
    ey = torch.floor(torch.log2(y.abs().clamp_min(2.0 ** emin))).clamp_min(emin)
    step = torch.exp2(ey - mbits)
    q = torch.round(y / step) * step          # torch.round: half to even
    q = q.clamp(-vmax, vmax)
    return (q * sc).reshape(shp)


def _q(x: Tensor) -> Tensor:
    """Hand-over rounding point: identity unless emulate_rounding() is active."""
    return x if _Emu.dtype is None else x.to(_Emu.dtype).to(torch.float32)


def _fp8_round(x: Tensor) -> Tensor:
    """Per-tensor power-of-two scale that puts amax in [112, 224], saturating e4m3 rounding, back to fp32 (the engine's static
    scales differ from this only in the exponent they pick, which does not change a floating-point format's relative error).
    With ``_Emu.fp8_block`` = 32 the scale is per 32 consecutive elements of the contraction (last) axis instead — the E8M0 block
    scale v_mfma_scale_f32_32x32x64_f8f6f4 applies per lane (tools/fp8_scale_study.py)."""
    if _Emu.fp8_block and x.shape[-1] % _Emu.fp8_block == 0:
        b = x.reshape(*x.shape[:-1], x.shape[-1] // _Emu.fp8_block, _Emu.fp8_block)
        amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
        sc = torch.exp2(torch.floor(torch.log2(224.0 / amax)))
        return ((b * sc).clamp(-448.0, 448.0).to(_Emu.operand_dtype).to(torch.float32) / sc).reshape(x.shape)
    amax = float(x.abs().max())
    if amax == 0.0:
        return x
    s = 2.0 ** math.floor(math.log2(224.0 / amax))
    return (x * s).clamp(-448.0, 448.0).to(_Emu.operand_dtype).to(torch.float32) / s


def _qa(x: Tensor, site: str = "") -> Tensor:
    """Hand-over point that is the A operand of a ViT / LLM layer linear (norm outputs, attention output, GELU / SwiGLU output)."""
    if site and site in _Emu.exact_sites:
        return x
    if site and _site_in(site, _Emu.lo_sites) and _Emu.dtype is not None and _Emu.operand_dtype is None:
        hi = _q(x)
        lo = _lo_round(x - hi, _Emu.lo_fmt, 32)
        if _Emu.lo_row_start > 0 and _Emu.tower == "llm":           # [..., S, D]: the rows below lo_row_start are handed over once (no image)
            lo = lo.clone()
            lo[..., :_Emu.lo_row_start, :] = 0
        return _Split(hi, lo)
    return _q(x) if _Emu.operand_dtype is None else _fp8_round(x)


def _site_in(site: str, sites) -> bool:
    """``site`` ("norm" / "attn_out" / "mlp_act") is selected by ``sites`` either plainly or qualified with the running tower ("llm.mlp_act").
    Round 6: the two norm operands of a layer can be named apart ("norm1" = the q|k|v operand, "norm2" = the fc1 / gate-up operand; "norm"
    still means both) and an entry may end in a layer range, "llm.norm2@8-31" (inclusive)."""
    names = (site, _Emu.sub) if (site == "norm" and _Emu.sub) else (site,)
    for n in names:
        if n in sites or (_Emu.tower + "." + n) in sites:
            return True
    for ent in sites:
        if "@" not in ent:
            continue
        key, rng = ent.split("@", 1)
        if key in names or any(key == _Emu.tower + "." + n for n in names):
            lo, _, hi = rng.partition("-")
            if int(lo) <= _Emu.layer <= int(hi or lo):
                return True
    return False


def _lin(h, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """A ViT / LLM layer linear.  A plain operand: F.linear on the (possibly fp8-rounded) weight.  A _Split operand: the 16-bit pass on
    the exact weight plus the low-bit pass — residual image x weight image — into the same sum."""
    if isinstance(h, _Split):                        # (the weight image is rebuilt per call: a cache of every layer's would double the oracle's memory)
        return F.linear(h.hi, w, b) + F.linear(h.lo, _lo_round(w, _Emu.lo_fmt, _Emu.lo_wblock))
    return F.linear(h, _wq(w), b)


def _wq(w: Tensor) -> Tensor:
    """Weight of a ViT / LLM layer linear: exact unless the fp8 schedule is emulated."""
    if _Emu.operand_dtype is None:
        return w
    hit = _Emu._wcache.get(id(w))
    if hit is None or hit[0] is not w:
        hit = _Emu._wcache[id(w)] = (w, _fp8_round(w))
    return hit[1]


def _tr(name: str, x: Tensor) -> None:
    if _Emu.trace is not None:
        _Emu.trace.append((name, x.detach().clone()))


class emulate_rounding:
    def __init__(self, dtype, trace: Optional[list] = None, operand_dtype=None, exact_sites=(), fp8_block: int = 0, fp8_attention: bool = False,
                 lo_sites=(), lo_fmt: str = "e2m1", lo_wblock: int = 0, lo_row_start: int = 0):
        self.dtype, self.trace, self.operand_dtype, self.exact_sites = dtype, trace, operand_dtype, frozenset(exact_sites)
        self.lo = (frozenset(lo_sites), lo_fmt, lo_wblock)
        self.lo_row_start = int(lo_row_start)
        self.fp8_block = fp8_block
        self.fp8_attention = bool(fp8_attention) and operand_dtype is not None

    def __enter__(self):
        self._old = (_Emu.dtype, _Emu.trace, _Emu.operand_dtype, _Emu.fused)
        self._old_sites, self._old_block, self._old_a8 = _Emu.exact_sites, _Emu.fp8_block, _Emu.fp8_attention
        self._old_lo = (_Emu.lo_sites, _Emu.lo_fmt, _Emu.lo_wblock)
        _Emu.lo_sites, _Emu.lo_fmt, _Emu.lo_wblock = self.lo
        self._old_row_start, _Emu.lo_row_start = _Emu.lo_row_start, self.lo_row_start
        _Emu.fp8_block = self.fp8_block
        _Emu.fp8_attention = self.fp8_attention
        _Emu.dtype, _Emu.trace, _Emu.operand_dtype = self.dtype, self.trace, self.operand_dtype
        _Emu.exact_sites = self.exact_sites
        if self.operand_dtype is not None:
            _Emu.fused = False               # the fp8 schedule keeps the norms as launches of their own (lmi_norm_fp8)
        return self

    def __exit__(self, *exc):
        _Emu.dtype, _Emu.trace, _Emu.operand_dtype, _Emu.fused = self._old
        _Emu.exact_sites, _Emu.fp8_block, _Emu.fp8_attention = self._old_sites, self._old_block, self._old_a8
        _Emu.lo_sites, _Emu.lo_fmt, _Emu.lo_wblock = self._old_lo
        _Emu.lo_row_start = self._old_row_start
        _Emu._wcache.clear()
        return False


# ==================================================================================================
# a1-a4  tiler (integers + PIL); kept as an independent restatement so tests can diff it against
#        leopard_amd.tiler AND against the reference-generated fixtures
# ==================================================================================================
def allocate_patches(image_sizes, patch_size=364, patch_budget=50):
    """EVAL:26-58."""
    counts = []
    for size in image_sizes:
        h, w = size                                   # sic: PIL gives (W,H); symmetric product
        n = round(h / patch_size) * round(w / patch_size)
        counts.append(0 if n == 1 else n)
    total = sum(counts)
    if total <= patch_budget:
        return counts
    f = patch_budget / total
    scaled = [int(c * f) for c in counts]
    while sum(scaled) > patch_budget:
        excess = sum(scaled) - patch_budget
        for i in range(len(scaled)):
            if scaled[i] > 0:
                scaled[i] -= 1
                excess -= 1
            if excess == 0:
                break
    return scaled


def select_best_resolution(original_size, num_patches, patch_size=364):
    """EVAL:61-99."""
    if num_patches == 0:
        return None
    ow, oh = original_size
    best, max_eff, min_waste = None, 0, float("inf")
    for row in range(1, num_patches + 1):
        for col in range(1, num_patches + 1):
            if row * col > num_patches or (row == 1 and col == 1):
                continue
            height, width = row * patch_size, col * patch_size
            scale = min(width / ow, height / oh)
            dw, dh = int(ow * scale), int(oh * scale)
            eff = min(dw * dh, ow * oh)
            waste = width * height - eff
            if eff > max_eff or (eff == max_eff and waste < min_waste):
                max_eff, min_waste, best = eff, waste, (width, height)
    if best == (patch_size, patch_size):
        return None
    return best


def resize_and_pad_image(image, target_resolution):
    """EVAL:102-140 (PIL default resample for Image.resize is BICUBIC)."""
    if target_resolution is None:
        return None
    from PIL import Image
    ow, oh = image.size
    tw, th = target_resolution
    sw, sh = tw / ow, th / oh
    if sw < sh:
        nw, nh = tw, min(math.ceil(oh * sw), th)
    else:
        nh, nw = th, min(math.ceil(ow * sh), tw)
    resized = image.resize((nw, nh))
    canvas = Image.new("RGB", (tw, th), (0, 0, 0))
    canvas.paste(resized, ((tw - nw) // 2, (th - nh) // 2))
    return canvas


def divide_to_patches(image, patch_size):
    """EVAL:143-162."""
    w, h = image.size
    out = []
    for i in range(0, h, patch_size):
        for j in range(0, w, patch_size):
            out.append(image.crop((j, i, j + patch_size, i + patch_size)))
    return out


def tile_sample(images, patch_size=364, sample_budget=50):
    """EVAL:386-401: returns (list of PIL ViT inputs, num_patchs_per_images_real)."""
    budget = sample_budget - len(images)
    if budget > 0:
        alloc = allocate_patches([im.size for im in images], patch_size, budget)
        res = [select_best_resolution(im.size, n, patch_size) for im, n in zip(images, alloc)]
        padded = [resize_and_pad_image(im, r) for im, r in zip(images, res)]
        patches = [divide_to_patches(p, patch_size) if p is not None else [] for p in padded]
        real = [len(p) for p in patches]
        allp = []
        for origin, p in zip(images, patches):
            allp += [origin] + p
        return allp, real
    return list(images), [1] * len(images)


def siglip_image_processor(image, size=364) -> Tensor:
    """SiglipImageProcessor.preprocess as called at EVAL:404 (third-party; config of
    siglip-so400m-14-364: resize to size x size BICUBIC, rescale 1/255, normalize mean=std=0.5)."""
    from PIL import Image
    im = image.convert("RGB")
    if im.size != (size, size):
        im = im.resize((size, size), resample=Image.BICUBIC)
    x = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).to(torch.float32)
    x = x * (1.0 / 255.0)
    x = (x - 0.5) / 0.5
    return x.permute(2, 0, 1).unsqueeze(0).contiguous()


# ==================================================================================================
# a7  SigLIP vision tower  (third-party SiglipVisionModel; in-tree analogue IVT:104-178)
# ==================================================================================================
def gelu_tanh(x: Tensor) -> Tensor:
    return F.gelu(x, approximate="tanh")


def _softmax_q(scores: Tensor, fp8_p: bool = False) -> Tensor:
    """fp32 softmax; under emulate_rounding() the probabilities that enter the P.V product are rounded like the
    kernel's P operand while the normaliser stays the fp32 sum of the UNROUNDED exponentials (as in the kernel).
    fp8_p (attention_fp8.h): that operand is e4m3 — exp(s - max) <= 1 rounded directly (the kernel's P is the same value times a power
    of two <= 2^8, its deferred reference: the same relative rounding, and fewer values flushed at the bottom of the range)."""
    if _Emu.dtype is None:
        return torch.softmax(scores, dim=-1, dtype=torch.float32)
    m = scores.amax(dim=-1, keepdim=True)
    e = torch.exp(scores - m)
    if fp8_p:
        return e.to(_Emu.operand_dtype).to(torch.float32) / e.sum(dim=-1, keepdim=True)
    return _q(e) / e.sum(dim=-1, keepdim=True)


def siglip_embeddings(pixel_values: Tensor, W: Dict[str, Tensor], cfg) -> Tensor:
    """patch conv (k=stride=patch, bias) -> flatten -> + position embedding.  IVT:57-64,118-150."""
    p = "vision_tower.vision_model.embeddings."
    x = F.conv2d(_q(pixel_values), W[p + "patch_embedding.weight"], W[p + "patch_embedding.bias"],
                 stride=cfg.vision_config.patch_size)
    x = x.flatten(2).transpose(1, 2)
    return x + W[p + "position_embedding.weight"].unsqueeze(0)


def siglip_layer(x: Tensor, W: Dict[str, Tensor], i: int, cfg, prefix: str = "vision_tower.vision_model.") -> Tensor:
    vc = cfg.vision_config
    p = f"{prefix}encoder.layers.{i}."
    N, T, D = x.shape
    H, hd = vc.num_attention_heads, vc.head_dim
    r = x
    h = _qa(F.layer_norm(x, (D,), W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], vc.layer_norm_eps), "norm")
    q = _q(_lin(h, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"])).view(N, T, H, hd).transpose(1, 2)
    k = _q(_lin(h, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"])).view(N, T, H, hd).transpose(1, 2)
    v = _q(_lin(h, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"])).view(N, T, H, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)      # full (non-causal) attention per tile
    a = _softmax_q(s)
    o = torch.matmul(a, v).transpose(1, 2).reshape(N, T, D)
    o = _qa(o if ("attn_out" in _Emu.exact_sites or _site_in("attn_out", _Emu.lo_sites)) else _q(o), "attn_out")
    x = r + _lin(o, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"])
    r = x
    h = _qa(F.layer_norm(x, (D,), W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], vc.layer_norm_eps), "norm")
    h = _qa(gelu_tanh(_lin(h, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"])), "mlp_act")
    return r + _lin(h, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"])


def siglip_vision_tower(pixel_values: Tensor, W: Dict[str, Tensor], cfg) -> Tensor:
    """``self.vision_tower(pixel_values).last_hidden_state`` (EVAL:268-273): [N,3,S,S] -> [N,T,D],
    post-layernorm included, pooling head not computed (its output is unused by the reference)."""
    x = siglip_embeddings(pixel_values, W, cfg)
    _tr("vit.embed", x)
    _Emu.tower = "vit"
    for i in range(cfg.vision_config.num_hidden_layers):
        _Emu.layer = i
        x = siglip_layer(x, W, i, cfg)
        _tr(f"vit.{i}", x)
    _Emu.tower, _Emu.layer = "", -1
    p = "vision_tower.vision_model.post_layernorm."
    return _q(F.layer_norm(x, (x.shape[-1],), W[p + "weight"], W[p + "bias"], cfg.vision_config.layer_norm_eps))


# ==================================================================================================
# a8-a9  pixel shuffle + projector   (EVAL:165-192; duplicate VLM:456-466)
# ==================================================================================================
def pixel_shuffle(x: Tensor, scale_factor: int = 2) -> Tensor:
    """EVAL:165-176 restated as the closed-form gather
    out[n, (h//2)*(G/2) + (w//2), (h%2)*2D + (w%2)*D + c] = x[n, h*G + w, c]   (G = sqrt(T))."""
    n, t, d = x.shape
    g = int(t ** 0.5)
    s = scale_factor
    x = x.view(n, g // s, s, g // s, s, d)          # [n, ph, dh, pw, dw, c]
    x = x.permute(0, 1, 3, 2, 4, 5)                 # [n, ph, pw, dh, dw, c]
    return x.reshape(n, (g // s) * (g // s), s * s * d)


def projector(image_features: Tensor, W: Dict[str, Tensor]) -> Tensor:
    """myLlavaMultiModalProjector.forward, EVAL:187-192 (act = ACT2FN['gelu'] = erf GELU)."""
    p = "multi_modal_projector."
    h = pixel_shuffle(image_features)
    h = _q(F.gelu(F.linear(h, W[p + "linear_1.weight"], W[p + "linear_1.bias"])))
    return F.linear(h, W[p + "linear_2.weight"], W[p + "linear_2.bias"])


# ==================================================================================================
# a10  embedding gather + image/text merge  (EVAL:263,284-287; third-party 4.38 merge; analogue
#      scatter VLM:526-533)
# ==================================================================================================
def merge_plan(input_ids: np.ndarray, image_token_index: int, n_features: int, tokens_per_tile: int) -> np.ndarray:
    """Index map of the merged sequence for ONE unpadded sample.

    Returns ``src`` of length S = S_in + (tokens_per_tile-1) * n_img_tokens with
    ``src[s] = t >= 0``  -> row s is the text embedding of input position t, and
    ``src[s] = -(j+1)``  -> row s is visual-feature row j (row-major over [N, tokens_per_tile]).
    Raises ValueError exactly when 4.38's merge does: #image tokens * tokens_per_tile != #feature rows."""
    ids = np.asarray(input_ids).reshape(-1)
    special = ids == image_token_index
    if int(special.sum()) * tokens_per_tile != n_features:
        raise ValueError(
            f"The input provided to the model are wrong. The number of image tokens is {int(special.sum())} "
            f"while the number of image given to the model is {n_features // max(tokens_per_tile, 1)}. "
            "This prevents correct indexing and breaks batch generation.")
    width = np.where(special, tokens_per_tile, 1)
    end = np.cumsum(width) - 1                        # last slot of each expanded input position
    S = int(end[-1]) + 1 if len(end) else 0
    src = np.empty(S, dtype=np.int64)
    feat = 0
    for t in range(len(ids)):
        if special[t]:
            a = end[t] - tokens_per_tile + 1
            src[a:end[t] + 1] = -(np.arange(feat, feat + tokens_per_tile) + 1)
            feat += tokens_per_tile
        else:
            src[end[t]] = t
    return src


def embed_and_merge(input_ids: Tensor, image_features: Tensor, W: Dict[str, Tensor], cfg) -> Tuple[Tensor, Tensor, Tensor]:
    """Returns (inputs_embeds[1,S,D], attention_mask[1,S], position_ids[1,S]) for one unpadded sample:
    mask is all ones and position_ids = cumsum(mask)-1 = arange(S)."""
    ids = input_ids.reshape(-1)
    emb = F.embedding(ids, W["language_model.model.embed_tokens.weight"])
    feats = image_features.reshape(-1, image_features.shape[-1])
    src = torch.from_numpy(merge_plan(ids.numpy(), cfg.image_token_index, feats.shape[0], image_features.shape[1]))
    out = torch.empty(src.numel(), emb.shape[-1], dtype=emb.dtype)
    is_text = src >= 0
    out[is_text] = emb[src[is_text]]
    out[~is_text] = feats[-src[~is_text] - 1].to(emb.dtype)
    S = src.numel()
    return out.unsqueeze(0), torch.ones(1, S, dtype=torch.long), torch.arange(S).unsqueeze(0)


# ==================================================================================================
# a11  Llama-3.1 decoder  (third-party LlamaForCausalLM; in-tree analogues XFMR:678-885 attention,
#      XFMR:97-176 SwiGLU MLP, XFMR:1208-1340 layer, RMS:26-31, ROPE:48-83,197-239)
# ==================================================================================================
def llama3_inv_freq(head_dim: int, theta: float, scaling) -> Tensor:
    """ROPE:48-83 (factor 8, low 1, high 4, original context 8192) on top of the plain 1/theta^(2i/d)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    if scaling is None:
        return inv
    low_wl = scaling.original_max_position_embeddings / scaling.low_freq_factor
    high_wl = scaling.original_max_position_embeddings / scaling.high_freq_factor
    wl = 2 * math.pi / inv
    out = torch.where(wl > low_wl, inv / scaling.factor, inv)
    smooth = (scaling.original_max_position_embeddings / wl - scaling.low_freq_factor) / (
        scaling.high_freq_factor - scaling.low_freq_factor)
    smoothed = (1 - smooth) * out / scaling.factor + smooth * out
    medium = ~(wl < high_wl) & ~(wl > low_wl)
    return torch.where(medium, smoothed, out)


def rope_tables(position_ids: Tensor, head_dim: int, theta: float, scaling) -> Tuple[Tensor, Tensor]:
    """cos/sin [S, head_dim] with the rotate-half layout cat(freqs, freqs).  ROPE:197-239."""
    inv = llama3_inv_freq(head_dim, theta, scaling)
    f = position_ids.reshape(-1, 1).to(torch.float32) * inv.reshape(1, -1)
    emb = torch.cat((f, f), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """RMS:26-31."""
    v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def _rms_norm_q(x: Tensor, w: Tensor, eps: float, first: bool, sub: str = "") -> Tensor:
    """rms_norm followed by the hand-over rounding of the HIP path (identity without emulate_rounding)."""
    _Emu.sub = sub
    try:
        return _rms_norm_q_impl(x, w, eps, first)
    finally:
        _Emu.sub = ""


def _rms_norm_q_impl(x: Tensor, w: Tensor, eps: float, first: bool) -> Tensor:
    if "norm" in _Emu.exact_sites:
        return rms_norm(x, w, eps)
    if _Emu.dtype is None or not _Emu.fused or first:
        return _qa(rms_norm(x, w, eps), "norm" if _site_in("norm", _Emu.lo_sites) else "")
    v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    if _site_in("norm", _Emu.lo_sites):              # the producer also emits the low-bit image of x * gamma - T(x * gamma)
        return _qa(x * w, "norm").scaled(torch.rsqrt(v + eps))
    return _q(x * w) * torch.rsqrt(v + eps)          # producer epilogue rounds x * gamma; the consumer applies rstd in fp32


def llama_layer(x: Tensor, W: Dict[str, Tensor], i: int, cfg, cos: Tensor, sin: Tensor,
                kv_out: Optional[list] = None, prefix: str = "language_model.model.") -> Tensor:
    tc = cfg.text_config
    p = f"{prefix}layers.{i}."
    B, S, D = x.shape
    H, KV, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    r = x
    pre = (lambda t: t) if _Emu.fused else _q          # unfused schedule: q / k are also rounded before the rotation
    h = _rms_norm_q(x, W[p + "input_layernorm.weight"], tc.rms_norm_eps, first=(i == 0), sub="norm1")
    q = pre(_lin(h, W[p + "self_attn.q_proj.weight"])).view(B, S, H, hd).transpose(1, 2)
    k = pre(_lin(h, W[p + "self_attn.k_proj.weight"])).view(B, S, KV, hd).transpose(1, 2)
    v = _q(_lin(h, W[p + "self_attn.v_proj.weight"])).view(B, S, KV, hd).transpose(1, 2)
    q = _q(q * cos + rotate_half(q) * sin)
    k = _q(k * cos + rotate_half(k) * sin)
    if kv_out is not None:
        kv_out.append((k, v))
    a8 = _Emu.fp8_attention and _Emu.operand_dtype is not None   # lmi_attn_prep_fp8: the attention reads e4m3 copies of the 16-bit q / k / v
    if a8:
        q, k, v = _fp8_round(q), _fp8_round(k), _fp8_round(v)
    rep = H // KV                                               # GQA repeat, XFMR:829-836
    kk = k.repeat_interleave(rep, dim=1)
    vv = v.repeat_interleave(rep, dim=1)
    o = torch.empty(B, H, S, hd, dtype=x.dtype)
    scale = hd ** -0.5
    step = 1024                                                 # query chunking only bounds memory
    ar = torch.arange(S)
    for s0 in range(0, S, step):
        s1 = min(S, s0 + step)
        sc = torch.matmul(q[:, :, s0:s1], kk[:, :, :s1].transpose(-1, -2)) * scale
        causal = ar[s0:s1, None] >= ar[None, :s1]
        if getattr(tc, "sliding_window", None):                 # Mistral: query i sees keys j with i - j < window
            causal = causal & (ar[s0:s1, None] - ar[None, :s1] < tc.sliding_window)
        sc = sc.masked_fill(~causal, float("-inf"))
        o[:, :, s0:s1] = torch.matmul(_softmax_q(sc, fp8_p=a8), vv[:, :, :s1])
    o = o.transpose(1, 2).reshape(B, S, H * hd)
    o = _qa(o if ("attn_out" in _Emu.exact_sites or _site_in("attn_out", _Emu.lo_sites)) else _q(o), "attn_out")
    x = r + _lin(o, W[p + "self_attn.o_proj.weight"])
    r = x
    h = _rms_norm_q(x, W[p + "post_attention_layernorm.weight"], tc.rms_norm_eps, first=False, sub="norm2")
    g = _lin(h, W[p + "mlp.gate_proj.weight"])
    u = _lin(h, W[p + "mlp.up_proj.weight"])
    return r + _lin(_qa(F.silu(g) * u, "mlp_act"), W[p + "mlp.down_proj.weight"])       # XFMR:136-139


def llama_forward(inputs_embeds: Tensor, position_ids: Tensor, W: Dict[str, Tensor], cfg,
                  last_only: bool = False, kv_out: Optional[list] = None, prefix: str = "language_model.model.",
                  head: str = "language_model.lm_head.weight") -> Tensor:
    """``self.language_model(inputs_embeds=..., position_ids=...)`` -> logits (EVAL:322-333).
    ``last_only`` computes the head for the final position only (the algorithmic need of prefill);
    the reference computes all positions."""
    tc = cfg.text_config
    cos, sin = rope_tables(position_ids, tc.head_dim, tc.rope_theta, tc.rope_scaling)
    x = inputs_embeds
    _tr("llm.embed", x)
    _Emu.tower = "llm"
    for i in range(tc.num_hidden_layers):
        _Emu.layer = i
        x = llama_layer(x, W, i, cfg, cos, sin, kv_out, prefix=prefix)
        _tr(f"llm.{i}", x)
    _Emu.tower, _Emu.layer = "", -1
    x = rms_norm(x, W[prefix + "norm.weight"], tc.rms_norm_eps)
    if last_only:
        x = x[:, -1:, :]
        if not _Emu.fp32_head:
            x = _q(x)
    else:
        x = _q(x)                      # the all-position head is an MFMA GEMM over 16-bit rows
    return F.linear(x, W[head])


# ==================================================================================================
# the whole prefill (EVAL:201-361, branch "pixel_values is not None and input_ids.shape[1] != 1")
# ==================================================================================================
@torch.no_grad()
def prefill_logits(input_ids: Tensor, pixel_values: Tensor, W: Dict[str, Tensor], cfg,
                   last_only: bool = False, return_parts: bool = False):
    feats = siglip_vision_tower(pixel_values, W, cfg)                  # EVAL:268-273
    vis = projector(feats, W)                                          # EVAL:283
    emb, mask, pos = embed_and_merge(input_ids, vis, W, cfg)           # EVAL:263,285-287
    logits = llama_forward(emb, pos, W, cfg, last_only=last_only)      # EVAL:322-333
    if return_parts:
        return logits, {"vit": feats, "visual_tokens": vis, "inputs_embeds": emb, "position_ids": pos}
    return logits


@torch.no_grad()
def greedy_generate(input_ids: Tensor, pixel_values: Tensor, W: Dict[str, Tensor], cfg,
                    max_new_tokens: int, eos_token_id: Sequence[int] = ()) -> Tensor:
    """Greedy decode restated WITHOUT a KV cache (recomputes the text suffix; only for tiny configs):
    EVAL:448-452 semantics — argmax, stop at eos, returns [1, S_in + T]."""
    feats = projector(siglip_vision_tower(pixel_values, W, cfg), W)
    ids = input_ids.reshape(1, -1).clone()
    for _ in range(max_new_tokens):
        emb, _, pos = embed_and_merge(ids, feats, W, cfg)
        nxt = int(llama_forward(emb, pos, W, cfg, last_only=True)[0, -1].argmax())
        ids = torch.cat([ids, torch.tensor([[nxt]])], dim=1)
        if nxt in set(eos_token_id):
            break
    return ids


def weights_from_numpy(sd: Dict[str, np.ndarray]) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.float32) for k, v in sd.items()}

"""Leopard-Idefics2 on the HIP kernels (BASELINE config 4).

Reference call site: evaluations/models/idefics2_multiimg.py ("IDEF") loads stock Idefics2 with
``do_image_splitting=False, longest_edge=980`` (IDEF:22-30) and calls ``model.generate(**inputs)`` (IDEF:95); all
arithmetic is third-party ``Idefics2ForConditionalGeneration``.  In-tree descriptions of the same math:
megatron_patch/model/idefics2/idefics_vision_tower.py:104-178 (NaViT tower), perceiver_transformer.py:582-706,1581-1722
(perceiver), idefics_vlm_model.py:563-645 (merger + forward).

Data path (every arithmetic step is a libleopard_amd.so call; the LLM core is LeopardEngine's):
    per image (native aspect ratio, no padding):  lmi_preprocess_images -> im2col rows
      -> patch GEMM (+bias + position table row picked by the NaViT bucketised position id, `add_rows`)
    all images packed as one varlen batch (cu_seqlens = patch counts; HF's padded + masked batch gives the same values):
      -> 27 x SigLIP layer (LN, QKV GEMM, non-causal varlen attention d=72, out GEMM(+res), LN, fc1(+gelu), fc2(+res)) -> post-LN
      -> modality projection: gate/up GEMM (+SwiGLU epilogue) -> down GEMM                       [P, 4096] fp32
      -> perceiver x3: RMSNorm(latents), RMSNorm(context) written into one packed [context_i ; latents_i] operand,
           q GEMM, fused k|v GEMM, cross attention (64 queries x (P_i + 64) keys, d=96, GQA 16/4), o GEMM(+res),
           RMSNorm, gate/up GEMM (+SwiGLU), down GEMM(+res) -> RMSNorm                           [n_img*64, 4096]
    -> lmi_embed_merge (1 feature row per <image> id) -> Mistral-7B prefill (theta 1e4, sliding window 4096) -> logits
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config import Idefics2Config
from .engine import KVCache, LeopardEngine, PrefillResult
from .ops import Ops
from .synth import KIND_WEIGHT, idefics2_param_specs, name_seed
from .weights import LlmLayerW, VitLayerW, _pad1, _pad2, _round_up, build_llm_layers, check_tp_degree, interleave_gate_up


class Idefics2SynthSource:
    def __init__(self, cfg: Idefics2Config, ops, device, dtype):
        self.ops, self.device, self.dtype = ops, device, dtype
        self.specs = {n: (s, k) for n, s, k in idefics2_param_specs(cfg)}

    def get(self, name: str) -> torch.Tensor:
        shape, kind = self.specs[name]
        out = torch.empty(shape, dtype=self.dtype if kind == KIND_WEIGHT else torch.float32, device=self.device)
        self.ops.fill_synthetic(out, name_seed(name), kind)
        return out


@dataclass
class PerceiverLayerW:
    lat_norm: torch.Tensor; ctx_norm: torch.Tensor; post_norm: torch.Tensor
    q_w: torch.Tensor; kv_w: torch.Tensor; o_w: torch.Tensor
    gu_w: torch.Tensor; down_w: torch.Tensor


@dataclass
class Idefics2Weights:
    cfg: Idefics2Config
    dtype: torch.dtype
    patch_w: torch.Tensor = None; patch_b: torch.Tensor = None; pos_emb: torch.Tensor = None
    vit_layers: List[VitLayerW] = field(default_factory=list)
    post_ln_w: torch.Tensor = None; post_ln_b: torch.Tensor = None
    mp_gu_w: torch.Tensor = None; mp_down_w: torch.Tensor = None
    latents: torch.Tensor = None
    perceiver_layers: List[PerceiverLayerW] = field(default_factory=list)
    perceiver_norm: torch.Tensor = None
    embed: torch.Tensor = None
    llm_layers: List[LlmLayerW] = field(default_factory=list)
    final_norm: torch.Tensor = None
    lm_head: torch.Tensor = None
    patch_k: int = 0; vit_ff: int = 0; llm_ff: int = 0; q_w_rows: int = 0; kv_rows: int = 0
    mp_ff: int = 0                           # modality-projection FFN width (never sharded; = the text model's intermediate size)
    # tensor-parallel shard of the Mistral decoder (BASELINE config 4: "TP=8 LLM over xGMI"); the vision side is replicated and
    # sharded by IMAGE at run time (Idefics2Engine.encode_images_sharded)
    llm_heads: int = 0; llm_kv_heads: int = 0; tp_rank: int = 0; tp_size: int = 1

    @classmethod
    def build(cls, cfg: Idefics2Config, source, dtype, tp_rank: int = 0, tp_size: int = 1) -> "Idefics2Weights":
        vc, tc, pc = cfg.vision_config, cfg.text_config, cfg.perceiver_config
        W = cls(cfg=cfg, dtype=dtype)
        check_tp_degree(tc, tp_size)
        W.tp_rank, W.tp_size = tp_rank, tp_size
        W.llm_heads, W.llm_kv_heads = tc.num_attention_heads // tp_size, tc.num_key_value_heads // tp_size
        g = source.get
        v = "model.vision_model."
        W.patch_k = _round_up(vc.patch_dim, 64)
        W.vit_ff = _round_up(vc.intermediate_size, 128)
        W.llm_ff = tc.intermediate_size // tp_size
        W.mp_ff = tc.intermediate_size
        W.patch_w = _pad2(g(v + "embeddings.patch_embedding.weight").reshape(vc.hidden_size, -1), vc.hidden_size, W.patch_k)
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
        c = "model.connector."
        W.mp_gu_w = interleave_gate_up(g(c + "modality_projection.gate_proj.weight"), g(c + "modality_projection.up_proj.weight"))
        W.mp_down_w = g(c + "modality_projection.down_proj.weight").contiguous()
        r = c + "perceiver_resampler."
        W.latents = g(r + "latents").to(torch.float32).contiguous()
        D = tc.hidden_size
        W.q_w_rows = _round_up(pc.n_heads * pc.head_dim, 128)
        W.kv_rows = _round_up(2 * pc.num_key_value_heads * pc.head_dim, 128)
        for i in range(pc.depth):
            p = f"{r}layers.{i}."
            kv = torch.cat([g(p + "self_attn.k_proj.weight"), g(p + "self_attn.v_proj.weight")], dim=0)
            o = g(p + "self_attn.o_proj.weight")
            W.perceiver_layers.append(PerceiverLayerW(
                lat_norm=g(p + "input_latents_norm.weight").float().contiguous(),
                ctx_norm=g(p + "input_context_norm.weight").float().contiguous(),
                post_norm=g(p + "post_attention_layernorm.weight").float().contiguous(),
                q_w=_pad2(g(p + "self_attn.q_proj.weight"), W.q_w_rows, D), kv_w=_pad2(kv, W.kv_rows, D),
                o_w=_pad2(o, D, _round_up(o.shape[1], 64)),
                gu_w=interleave_gate_up(g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")),
                down_w=g(p + "mlp.down_proj.weight").contiguous()))
        W.perceiver_norm = g(r + "norm.weight").float().contiguous()
        l = "model.text_model."
        W.embed = g(l + "embed_tokens.weight").contiguous()
        W.llm_layers = build_llm_layers(g, l, tc, tp_rank, tp_size)
        W.final_norm = g(l + "norm.weight").float().contiguous()
        head = g("lm_head.weight")
        W.lm_head = _pad2(head, _round_up(head.shape[0], 128), tc.hidden_size)
        return W


def navit_position_ids(nh: int, nw: int, grid: int) -> np.ndarray:
    """Bucketised fractional patch coordinates over the grid x grid position table (third-party
    Idefics2VisionEmbeddings.forward; idefics_vision_tower.py:118-150).  fp32 arithmetic, as the reference."""
    bounds = torch.arange(1 / grid, 1.0, 1 / grid).numpy()                          # the reference's fp32 boundaries
    fh = np.minimum(np.arange(nh, dtype=np.float32) * (np.float32(1.0) / np.float32(nh)), np.float32(1.0 - 1e-6))
    fw = np.minimum(np.arange(nw, dtype=np.float32) * (np.float32(1.0) / np.float32(nw)), np.float32(1.0 - 1e-6))
    bh = np.searchsorted(bounds, fh, side="right")
    bw = np.searchsorted(bounds, fw, side="right")
    return (bh[:, None] * grid + bw[None, :]).reshape(-1).astype(np.int32)


class Idefics2Engine(LeopardEngine):
    """LeopardEngine's LLM core (llm_prefill / decode_step / generate / embed_merge) + the Idefics2 vision side."""

    def __init__(self, cfg: Idefics2Config, weights: Idefics2Weights, ops: Optional[Ops] = None, device=None, use_tr: bool = True):
        super().__init__(cfg, weights, ops=ops, device=device, use_tr=use_tr)
        self.lo4_vit = True            # this model's lo4 mode was qualified (C4 full depth: 5.4e-4) with the NaViT tower corrected as well

    # ---- NaViT tower over a list of images of arbitrary sizes ------------------------------------------------
    def vision_tower_images(self, images: Sequence[torch.Tensor]):
        """images: per image u8 [H,W,3] or fp32 [3,H,W] (already resized by the processor rule).  Returns
        (post-LN features T [P, Dv], patch counts)."""
        ops, W, vc = self.ops, self.W, self.cfg.vision_config
        P_sz, D, H, hd = vc.patch_size, vc.hidden_size, vc.num_attention_heads, vc.head_dim
        counts, pos = [], []
        for im in images:
            ih, iw = (im.shape[0], im.shape[1]) if im.dtype == torch.uint8 else (im.shape[1], im.shape[2])
            nh, nw = ih // P_sz, iw // P_sz
            counts.append(nh * nw)
            pos.append(navit_position_ids(nh, nw, vc.grid))
        M = sum(counts)
        patches = self._empty(M, W.patch_k)
        off = 0
        for im, n in zip(images, counts):
            ops.preprocess_images(im.to(self.device).unsqueeze(0).contiguous(), patches[off:off + n], P_sz)
            off += n
        pos_dev = self._pinned_to_device(torch.from_numpy(np.concatenate(pos)))
        x = self._empty(M, D, dtype=torch.float32)
        ops.gemm(patches, W.patch_w, x, bias=W.patch_b, addmat=W.pos_emb, add_rows=pos_dev, epilogue=_lib.EPI_STORE_F32)
        cu_list = [0]
        for n in counts:
            cu_list.append(cu_list[-1] + n)
        cu = self._pinned_to_device(torch.tensor(cu_list, dtype=torch.int32))
        h = self._empty(M, D)
        qkv = self._empty(M, W.vit_layers[0].qkv_w.shape[0])
        if self.lo4 and self.lo4_vit:
            # precision "lo4" (LeopardEngine._vit_layers_lo4): every layer-linear operand of the NaViT tower travels with the fp4 image of its
            # rounding residual; the connector (modality projection + 3 perceiver layers) keeps the fast schedule, the Mistral layers take
            # LeopardEngine._llm_layers_lo4
            h4, att4, ff4 = self._lo4_act(M, D), self._lo4_act(M, D, heads=(H, hd)), self._lo4_act(M, W.vit_ff)
            for L, (qkv4, o4, fc14, fc24) in zip(W.vit_layers, self._lo4_weights("vit")):
                ops.norm_lo4(x, L.ln1_w, L.ln1_b, h4, vc.layer_norm_eps)
                ops.gemm_lo4(h4, L.qkv_w, qkv4, qkv, bias=L.qkv_b)
                ops.attention_lo4(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], att4, cu, cu, max(counts), H, H, hd, hd ** -0.5, False)
                ops.gemm_lo4(att4, L.o_w, o4, x, bias=L.o_b, epilogue=_lib.EPI_RESIDUAL)
                ops.norm_lo4(x, L.ln2_w, L.ln2_b, h4, vc.layer_norm_eps)
                ops.gemm_lo4(h4, L.fc1_w, fc14, ff4.hi, bias=L.fc1_b, act=_lib.ACT_GELU_TANH, out4=ff4)
                ops.gemm_lo4(ff4, L.fc2_w, fc24, x, bias=L.fc2_b, epilogue=_lib.EPI_RESIDUAL)
            ops.layernorm(x, W.post_ln_w, W.post_ln_b, h, vc.layer_norm_eps)
            return h, counts
        att = self._empty(M, D)
        ff = self._empty(M, W.vit_ff)
        for L in W.vit_layers:
            ops.layernorm(x, L.ln1_w, L.ln1_b, h, vc.layer_norm_eps)
            ops.gemm(h, L.qkv_w, qkv, bias=L.qkv_b)
            ops.attention(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], att, cu, cu, max(counts), H, H, hd, hd ** -0.5,
                          False, self.use_tr)
            ops.gemm(att, L.o_w, x, bias=L.o_b, epilogue=_lib.EPI_RESIDUAL)
            ops.layernorm(x, L.ln2_w, L.ln2_b, h, vc.layer_norm_eps)
            ops.gemm(h, L.fc1_w, ff, bias=L.fc1_b, act=_lib.ACT_GELU_TANH)
            ops.gemm(ff, L.fc2_w, x, bias=L.fc2_b, epilogue=_lib.EPI_RESIDUAL)
        ops.layernorm(x, W.post_ln_w, W.post_ln_b, h, vc.layer_norm_eps)
        return h, counts

    # ---- connector: modality projection + perceiver resampler --------------------------------------------------
    def connector(self, feats: torch.Tensor, counts: Sequence[int]) -> torch.Tensor:
        ops, W, cfg = self.ops, self.W, self.cfg
        tc, pc = cfg.text_config, cfg.perceiver_config
        D, Lt, n_img = tc.hidden_size, pc.n_latents, len(counts)
        P = feats.shape[0]
        gu = self._empty(P, W.mp_ff)
        ops.gemm(feats, W.mp_gu_w, gu, epilogue=_lib.EPI_SWIGLU)
        ctx = self._empty(P, D, dtype=torch.float32)
        ops.gemm(gu, W.mp_down_w, ctx, epilogue=_lib.EPI_STORE_F32)
        del gu
        lat = W.latents.repeat(n_img, 1).contiguous()                        # fp32 latent stream [n_img*Lt, D]
        hid = self._empty(P + n_img * Lt, D)                                 # packed [context_i ; latents_i] operand
        ln = self._empty(n_img * Lt, D)
        q = self._empty(n_img * Lt, W.q_w_rows)
        kv = self._empty(P + n_img * Lt, W.kv_rows)
        qd, kd = pc.n_heads * pc.head_dim, pc.num_key_value_heads * pc.head_dim
        att = self._empty(n_img * Lt, W.perceiver_layers[0].o_w.shape[1]) if W.perceiver_layers else None
        if att is not None and att.shape[1] > qd:
            att.zero_()                                                      # K padding columns of o_proj's operand
        g2 = self._empty(n_img * Lt, 4 * D)
        cu_q = [Lt * i for i in range(n_img + 1)]
        cu_k, hid_ctx, hid_lat, off, coff = [0], [], [], 0, 0
        for i, n in enumerate(counts):
            hid_ctx.append((coff, off, n))
            hid_lat.append((off + n, i * Lt))
            off += n + Lt
            coff += n
            cu_k.append(off)
        cu_q_t = self._pinned_to_device(torch.tensor(cu_q, dtype=torch.int32))
        cu_k_t = self._pinned_to_device(torch.tensor(cu_k, dtype=torch.int32))
        for L in W.perceiver_layers:
            ops.rmsnorm(lat, L.lat_norm, ln, pc.rms_norm_eps)
            for (c0, h0, n), (l0, q0) in zip(hid_ctx, hid_lat):
                ops.rmsnorm(ctx[c0:c0 + n], L.ctx_norm, hid[h0:h0 + n], pc.rms_norm_eps)
                hid[l0:l0 + Lt].copy_(ln[q0:q0 + Lt])                        # plumbing copy of 64 rows per image
            ops.gemm(ln, L.q_w, q)
            ops.gemm(hid, L.kv_w, kv)
            ops.attention(q[:, :qd], kv[:, :kd], kv[:, kd:2 * kd], att[:, :qd], cu_q_t, cu_k_t, Lt, pc.n_heads,
                          pc.num_key_value_heads, pc.head_dim, pc.head_dim ** -0.5, False, self.use_tr)
            ops.gemm(att, L.o_w, lat, epilogue=_lib.EPI_RESIDUAL)
            ops.rmsnorm(lat, L.post_norm, ln, pc.rms_norm_eps)
            ops.gemm(ln, L.gu_w, g2, epilogue=_lib.EPI_SWIGLU)
            ops.gemm(g2, L.down_w, lat, epilogue=_lib.EPI_RESIDUAL)
        out = self._empty(n_img * Lt, D, dtype=torch.float32)
        ops.rmsnorm(lat, W.perceiver_norm, out, pc.rms_norm_eps)             # fp32 in, fp32 out: the merged stream is fp32
        return out

    def encode_images(self, images: Sequence[torch.Tensor]) -> torch.Tensor:
        feats, counts = self.vision_tower_images(images)
        return self.connector(feats, counts)

    def encode_images_sharded(self, images: Sequence[torch.Tensor]) -> torch.Tensor:
        """Tensor-parallel runs: the images of the sample are independent through the tower, the modality projection and the
        perceiver, so rank r encodes images r, r + R, ... and ONE all-gather of the [n_latents, D] results restores the full set
        on every rank, in image order (bit-identical to ``encode_images``: no arithmetic crosses an image boundary)."""
        comm = self.comm
        n, R = len(images), comm.world
        Lt, D = self.cfg.perceiver_config.n_latents, self.cfg.text_config.hidden_size
        per = -(-n // R)
        mine = torch.zeros(per * Lt, D, dtype=torch.float32, device=self.device)
        own = [images[i] for i in range(comm.rank, n, R)]
        if own:
            mine[:len(own) * Lt] = self.encode_images(own)
        gathered = self._empty(R * per * Lt, D, dtype=torch.float32)
        comm.all_gather(gathered, mine)
        g = gathered.view(R, per, Lt, D)
        return torch.cat([g[i % R, i // R] for i in range(n)], dim=0)

    # ---- whole prefill (IDEF:91-95) ------------------------------------------------------------------------------
    @torch.no_grad()
    def prefill(self, input_ids: torch.Tensor, images: Optional[Sequence[torch.Tensor]], cache: Optional[KVCache] = None,
                all_logits: bool = False, keep_parts: bool = False, visual_tokens: Optional[torch.Tensor] = None) -> PrefillResult:
        if self.tp_size > 1:                                  # BASELINE config 4: one sample on all ranks
            if all_logits or keep_parts:
                raise NotImplementedError("all_logits / keep_parts are single-rank diagnostics")
            if visual_tokens is None and images is not None and len(images):
                visual_tokens = self.encode_images_sharded(images)
            return self._prefill_tp(input_ids, None, cache, visual_tokens)
        parts = {} if keep_parts else None
        if visual_tokens is None and images is not None and len(images):
            visual_tokens = self.encode_images(images)
        if keep_parts and visual_tokens is not None:
            parts["image_features"] = visual_tokens
        x = self.embed_merge(input_ids, visual_tokens)
        if keep_parts:
            parts["inputs_embeds"] = x.clone()
        S = x.shape[0]
        last, all_ = self.llm_prefill(x, [S], cache=cache, all_logits=all_logits)
        n_img = 0 if visual_tokens is None else visual_tokens.shape[0] // self.cfg.perceiver_config.n_latents
        return PrefillResult(logits_last=last[0], seq_len=S, n_tiles=n_img, logits_all=all_, parts=parts)


def resize_output_size(height: int, width: int, longest_edge: int) -> tuple:
    """Idefics2 processor size rule with shortest_edge = 0 (IDEF:23-25; third-party get_resize_output_image_size):
    shrink so that the longer side equals longest_edge; never enlarge."""
    ar = width / height
    if width >= height and width > longest_edge:
        width = longest_edge
        height = int(width / ar)
    elif height > width and height > longest_edge:
        height = longest_edge
        width = int(height * ar)
    return height, width


def preprocess_image_u8(image, longest_edge: int = 980) -> np.ndarray:
    """PIL image -> u8 [H,W,3] at the processor's output size (bilinear); rescale + normalise happen on the GPU."""
    from PIL import Image
    im = image.convert("RGB")
    h, w = resize_output_size(im.size[1], im.size[0], longest_edge)
    if (w, h) != im.size:
        im = im.resize((w, h), resample=Image.BILINEAR)
    return np.asarray(im, dtype=np.uint8)

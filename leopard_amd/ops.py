"""Thin torch-tensor front end over the C ABI (raw pointers, explicit shapes, current HIP stream).

PyTorch is used for device memory and streams only; every arithmetic step below is a call into
libleopard_amd.so.  ``Ops(lib)`` takes the ctypes handle so that the CPU kernel-logic emulator build
(tools/hipemu, tests only) can be driven through the very same wrappers with host tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (A_PIXEL_SHUFFLE, A_PLAIN, ACT_GELU_ERF, ACT_GELU_TANH, ACT_NONE, EPI_RESIDUAL, EPI_STORE,
                   EPI_STORE_F32, EPI_SWIGLU, LMI_BF16, LMI_F16, LMI_F32)

_DT = {torch.float16: LMI_F16, torch.bfloat16: LMI_BF16, torch.float32: LMI_F32}


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _ptr_f32(t: Optional[torch.Tensor], what: str = "bias"):
    """The C-ABI takes biases / addmats as ``const float*``: a 16-bit tensor would be read past its end as garbage, silently."""
    if t is not None and t.dtype != torch.float32:
        raise TypeError(f"leopard_amd: {what} must be float32 (the C-ABI reads const float*), got {t.dtype}")
    return _ptr(t)


def lo4_k4(K: int) -> int:
    """Width of the fp4 images of a K-wide operand: K rounded up to the 256-element k-tile of the correction phase."""
    return (K + 255) // 256 * 256


def lo4_head_k4(n_heads: int, head_dim: int) -> int:
    """Width of the per-head padded k order of an attention output's residual image (lmi_attn_varlen_fwd_lo4)."""
    return lo4_k4(n_heads * ((head_dim + 31) // 32) * 32)


class Lo4Act:
    """One A operand of a GEMM with the low-bit correction phase (include/leopard_amd.h ``lmi_lo4``): ``hi`` = T(x) [M, K], ``img`` = the
    fp4 (e2m1) image of x - T(x) [M, K4 / 2] bytes, ``sc`` = its E8M0 block scales [M, K4 / 32].  The three are views into caller-owned
    scratch; the producers (lmi_norm_lo4, lmi_split_lo4, the GEMM epilogues) write all of them, padding included."""
    __slots__ = ("hi", "img", "sc", "K", "K4", "buf", "row_sel", "unit_sel", "sel_ranges")

    def __init__(self, hi: torch.Tensor, img: torch.Tensor, sc: torch.Tensor):
        self.hi, self.img, self.sc, self.buf = hi, img, sc, None
        # row selection of the correction phase (lmi_lo4.row_sel / unit_sel; LeopardEngine.lo4_rows): uint8 [M] / [ceil(M / 64)] device tensors, or
        # None = every row.  With a selection the image and scales MUST start out zero (Lo4Act.empty(..., zero=True)): producers skip unselected rows
        # sel_ranges: optional HOST int32 numpy array [n, 2] of the selected [begin, end) row ranges (lmi_lo4.sel_ranges: a tile-order hint)
        self.row_sel = self.unit_sel = self.sel_ranges = None
        self.K = hi.shape[1]
        self.K4 = img.shape[1] * 2          # lo4_k4(K), or wider when the image carries a padded k order (attention outputs, head_dim 72 / 96)
        assert img.dtype == torch.uint8 and sc.dtype == torch.uint8 and self.K4 % 256 == 0 and self.K4 >= self.K and sc.shape[0] == hi.shape[0] == img.shape[0]
        assert sc.shape[1] >= self.K4 // 32 and sc.stride(0) % 4 == 0 and img.stride(0) % 16 == 0

    @staticmethod
    def empty(M: int, K: int, dtype, device, k4: Optional[int] = None, sel: Optional[tuple] = None) -> "Lo4Act":
        # a padded k order, or K not a multiple of 256: the attention kernel and the GEMM epilogues write their own blocks only — the padding
        # must read as zero codes / zero scales (the norm and split kernels write theirs); a row selection: the unselected rows are never written
        k4 = k4 or lo4_k4(K)
        alloc = torch.zeros if (k4 != K or sel is not None) else torch.empty
        act = Lo4Act(torch.empty(M, K, dtype=dtype, device=device), alloc(M, k4 // 2, dtype=torch.uint8, device=device),
                     alloc(M, k4 // 32, dtype=torch.uint8, device=device))
        if sel is not None:
            act.row_sel, act.unit_sel = sel[0], sel[1]
            act.sel_ranges = sel[2] if len(sel) > 2 else None
            assert act.row_sel.dtype == torch.uint8 and act.row_sel.numel() == M and act.unit_sel.numel() == (M + 63) // 64
        return act


def lo4_packed_act(M: int, K: int, dtype, device) -> Lo4Act:
    """A Lo4Act whose image and scales are column ranges of ONE row-major uint8 buffer ``buf`` [M, K4 / 2 + K4 / 32 (+ pad to 16)]: a
    tensor-parallel all-gather then moves both with one collective (rows are gathered rank-major, so row blocks stay row blocks)."""
    k4 = lo4_k4(K)
    rb = (k4 // 2 + k4 // 32 + 15) // 16 * 16
    buf = (torch.zeros if k4 != K else torch.empty)(M, rb, dtype=torch.uint8, device=device)
    act = Lo4Act(torch.empty(M, K, dtype=dtype, device=device), buf[:, :k4 // 2], buf[:, k4 // 2:k4 // 2 + k4 // 32])
    act.buf = buf
    return act


class Lo4Weight:
    """fp4 image of a weight [N, K4 / 2] bytes (row-major, whatever the layout of the 16-bit copy) + one E8M0 scale per row [N]."""
    __slots__ = ("img", "sc")

    def __init__(self, img: torch.Tensor, sc: torch.Tensor):
        self.img, self.sc = img, sc


class Ops:
    def __init__(self, lib=None, emulated: bool = False):
        self.lib = lib if lib is not None else _lib.load()
        self.emulated = emulated

    # ------------------------------------------------------------------------------------------
    def _stream(self, ref: torch.Tensor):
        if self.emulated:
            return C.c_void_p(0)
        if not ref.is_cuda:
            raise RuntimeError("leopard_amd ops need device (HIP) tensors; there is no CPU path")
        return C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)

    def _check(self, rc: int):
        if rc != 0:
            raise RuntimeError(f"libleopard_amd error {rc}: {self.lib.lmi_last_error().decode()}")

    def set_option(self, key: str, value: int):
        self._check(self.lib.lmi_set_option(key.encode(), int(value)))

    # ------------------------------------------------------------------------------------------
    def fill_synthetic(self, out: torch.Tensor, seed: int, kind: int):
        assert out.is_contiguous()
        self._check(self.lib.lmi_fill_synthetic(_ptr(out), out.numel(), seed & 0xFFFFFFFF, kind, _DT[out.dtype],
                                                self._stream(out)))
        return out

    def preprocess_tiles(self, src: torch.Tensor, out: torch.Tensor, image_size: int, patch: int):
        """src: u8 [N,S,S,3] or fp32 [N,3,S,S]; out: T [N*(S/P)^2, ldo]."""
        from_u8 = src.dtype == torch.uint8
        assert src.is_contiguous() and out.is_contiguous() and (from_u8 or src.dtype == torch.float32)
        self._check(self.lib.lmi_preprocess_tiles(_ptr(src), int(from_u8), _ptr(out), src.shape[0], image_size, patch,
                                                  out.shape[1], _DT[out.dtype], self._stream(out)))
        return out

    def patch_embed(self, pixels: torch.Tensor, w_fused: torch.Tensor, bias: torch.Tensor, pos_emb: torch.Tensor, out: torch.Tensor,
                    image_size: int, patch: int):
        """out fp32 [n*(S/P)^2, N] = conv(normalise(pixels)) + bias + pos_emb in ONE im2col + MFMA GEMM (lmi_patch_embed).
        pixels: u8 [n,S,S,3] or fp32 [n,3,S,S]; w_fused: T [N, KP] in image K order (weights.patch_weight_image_order)."""
        from_u8 = pixels.dtype == torch.uint8
        assert pixels.is_contiguous() and out.is_contiguous() and (from_u8 or pixels.dtype == torch.float32) and out.dtype == torch.float32
        self._check(self.lib.lmi_patch_embed(_ptr(pixels), int(from_u8), _ptr(w_fused), _ptr_f32(bias), _ptr(pos_emb), _ptr(out), pixels.shape[0],
                                             image_size, patch, w_fused.shape[0], w_fused.stride(0), out.stride(0), _DT[w_fused.dtype],
                                             self._stream(out)))
        return out

    def kv_append(self, k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, pos0: int):
        """cache rows [pos0, pos0 + S) = k / v rows (already rotated K; lmi_kv_append)."""
        S, width = k.shape
        if S == 0:
            return
        assert v.shape == k.shape and k.stride(1) == 1 and v.stride(1) == 1 and k.stride(0) == v.stride(0)
        assert k_cache.stride(0) == v_cache.stride(0) and pos0 + S <= k_cache.shape[0]
        self._check(self.lib.lmi_kv_append(_ptr(k), _ptr(v), _ptr(k_cache), _ptr(v_cache), S, width, k.stride(0), k_cache.stride(0), int(pos0),
                                           _DT[k.dtype], self._stream(k)))

    def gemm_bias_act(self, a, w, out, bias=None, act=ACT_NONE, residual=False, ps_grid=0, M=None):
        """SURVEY.md 8(b)'s short form of the linear (lmi_gemm_bias_act)."""
        N, K = w.shape
        if M is None:
            M = a.shape[0]
        self._check(self.lib.lmi_gemm_bias_act(_ptr(a), _ptr(w), _ptr(out), _ptr_f32(bias), M, N, K, a.stride(0), w.stride(0), out.stride(0), act,
                                               int(bool(residual)), int(ps_grid), _DT[w.dtype], self._stream(out)))
        return out

    def preprocess_images(self, src: torch.Tensor, out: torch.Tensor, patch: int):
        """src: u8 [n,H,W,3] or fp32 [n,3,H,W] (one size); out: T [n*(H//P)*(W//P), ldo]."""
        from_u8 = src.dtype == torch.uint8
        n, H, W = (src.shape[0], src.shape[1], src.shape[2]) if from_u8 else (src.shape[0], src.shape[2], src.shape[3])
        assert src.is_contiguous() and out.is_contiguous()
        self._check(self.lib.lmi_preprocess_images(_ptr(src), int(from_u8), _ptr(out), n, H, W, patch, out.shape[1],
                                                   _DT[out.dtype], self._stream(out)))
        return out

    def resample_u8(self, src: torch.Tensor, dst: torch.Tensor, axis: int, bounds: torch.Tensor, taps: torch.Tensor):
        """One pass of Pillow's 8-bit RGB resampling.  src/dst: u8 [rows, cols, 3] views whose last two dims are dense
        (dst may be a window of a larger canvas); bounds int32 [out, 2], taps int32 [out, ksize]."""
        assert src.dtype == torch.uint8 and dst.dtype == torch.uint8 and src.stride(2) == 1 and src.stride(1) == 3
        assert dst.stride(2) == 1 and dst.stride(1) == 3 and bounds.dtype == torch.int32 and taps.dtype == torch.int32
        assert bounds.is_contiguous() and taps.is_contiguous() and bounds.shape[0] == dst.shape[1 if axis == 0 else 0]
        self._check(self.lib.lmi_resample_u8(_ptr(src), _ptr(dst), axis, dst.shape[0], dst.shape[1], src.stride(0), dst.stride(0),
                                             _ptr(bounds), _ptr(taps), taps.shape[1], self._stream(dst)))
        return dst

    def layernorm(self, x, w, b, out, eps):
        M, D = x.shape
        self._check(self.lib.lmi_layernorm(_ptr(x), _ptr(w), _ptr(b), _ptr(out), M, D, x.stride(0), out.stride(0),
                                           float(eps), _DT[out.dtype], self._stream(out)))
        return out

    def rmsnorm(self, x, w, out, eps):
        M, D = x.shape
        self._check(self.lib.lmi_rmsnorm(_ptr(x), _ptr(w), _ptr(out), M, D, x.stride(0), out.stride(0), float(eps),
                                         _DT[out.dtype], self._stream(out)))
        return out

    @staticmethod
    def _ldw(w):
        """Leading dimension of a weight for the GEMM entries: LMI_LDW_PACKED(K) = -K when it is stored in the packed order
        (weights.mark_packed), its row stride otherwise."""
        return -w.shape[1] if getattr(w, "_lmi_packed", False) else w.stride(0)

    def gemm(self, a, w, out, bias=None, addmat=None, row_map=None, epilogue=EPI_STORE, act=ACT_NONE,
             a_mode=A_PLAIN, ps_grid=0, M=None, add_rows=None):
        """out = epilogue(a @ w.T).  a: T [M,K] (or the ViT output for pixel-shuffle mode), w: T [N,K]."""
        N, K = w.shape
        if M is None:
            M = a.shape[0]
        add_period = 0 if addmat is None else addmat.shape[0]
        self._check(self.lib.lmi_gemm(_ptr(a), _ptr(w), _ptr(out), _ptr_f32(bias), _ptr_f32(addmat, "addmat"), _ptr(add_rows), _ptr(row_map), M, N, K,
                                      a.stride(0), self._ldw(w), out.stride(0), add_period, epilogue, act, a_mode,
                                      ps_grid, _DT[w.dtype], self._stream(out)))
        return out

    def attention(self, q, k, v, out, cu_q, cu_k, max_seqlen_q, n_heads, n_kv_heads, head_dim, scale, causal,
                  use_tr=True, window=0):
        """q/k/v/out are 2-D row views [rows, >= heads*head_dim] (possibly column slices of a packed qkv buffer)."""
        n_seq = cu_q.numel() - 1
        self._check(self.lib.lmi_attn_varlen_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(cu_q), _ptr(cu_k), n_seq,
                                                 int(max_seqlen_q), n_heads, n_kv_heads, head_dim, q.stride(0),
                                                 k.stride(0), v.stride(0), out.stride(0), float(scale), int(causal),
                                                 int(window), int(use_tr), _DT[q.dtype], self._stream(out)))
        return out

    def attention_decode(self, q, k, v, out, cu_q, cu_k, max_seqlen_q, max_seqlen_k, n_heads, n_kv_heads, head_dim, scale,
                         workspace: torch.Tensor, window=0, hl=False):
        """Split-KV attention for a few query rows against a long cache (decode).  workspace: fp32, decode_workspace_elems().
        ``hl``: out has 2 x q rows — the output rows and, below them, the 16-bit residuals of their rounding (lmi_attn_decode_fwd_hl)."""
        n_seq = cu_q.numel() - 1
        assert not hl or out.shape[0] == 2 * q.shape[0]
        self._check((self.lib.lmi_attn_decode_fwd_hl if hl else self.lib.lmi_attn_decode_fwd)(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(cu_q), _ptr(cu_k), n_seq, int(max_seqlen_q),
                                                 int(max_seqlen_k), q.shape[0], n_heads, n_kv_heads, head_dim, q.stride(0), k.stride(0),
                                                 v.stride(0), out.stride(0), float(scale), int(window), _ptr(workspace),
                                                 workspace.numel() * workspace.element_size(), _DT[q.dtype], self._stream(out)))
        return out

    def attention_decode_pool(self, q, k, v, out, cu_q, k_begin, k_len, max_seqlen_k, n_heads, n_kv_heads, head_dim, scale,
                              workspace: torch.Tensor, window=0, hl=False):
        """Split-KV decode attention for a batch whose caches share one pooled buffer: sequence s owns rows
        [k_begin[s], k_begin[s] + k_len[s]) of k / v (int32 device tensors; k_len advances on the device).  ``hl``: as attention_decode."""
        n_seq = cu_q.numel() - 1
        assert not hl or out.shape[0] == 2 * q.shape[0]
        self._check((self.lib.lmi_attn_decode_pool_hl if hl else self.lib.lmi_attn_decode_pool)(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(cu_q), _ptr(k_begin), _ptr(k_len), n_seq, 1,
                                                  int(max_seqlen_k), q.shape[0], n_heads, n_kv_heads, head_dim, q.stride(0), k.stride(0),
                                                  v.stride(0), out.stride(0), float(scale), int(window), _ptr(workspace),
                                                  workspace.numel() * workspace.element_size(), _DT[q.dtype], self._stream(out)))
        return out

    def rope_qk_rows(self, qkv, n_q_heads, n_kv_heads, head_dim, cos_all, sin_all, k_cache, v_cache, cache_stride, pos_rows):
        """Batched decode: row s rotated at position pos_rows[s] (int32, device); K / V appended at row s * cache_stride + pos_rows[s]."""
        self._check(self.lib.lmi_rope_qk_rows(_ptr(qkv), qkv.shape[0], qkv.stride(0), n_q_heads, n_kv_heads, head_dim, _ptr(cos_all),
                                              _ptr(sin_all), _ptr(k_cache), _ptr(v_cache), k_cache.stride(0), int(cache_stride),
                                              _ptr(pos_rows), _DT[qkv.dtype], self._stream(qkv)))
        return qkv

    def gemm_skinny(self, w, x, out, epilogue=0, packed=None, rowsq_in=None, norm_dim=0, norm_eps=0.0, norm_out=None, norm_gamma=None, rowsq_out=None,
                    hl=False):
        """out[M <= 16, .] = epilogue(x @ w.T): the projections of a batched decode step.  epilogue: 0 store T, 1 fp32 +=,
        2 SwiGLU (w rows interleaved [32 gate | 32 up], out [M, N/2]), 3 store fp32.  packed: w is weights.skinny_pack(w) (same shape).
        rowsq_in / norm_out + norm_gamma + rowsq_out: the folded RMSNorm of lmi_gemm_skinny_ex (consumer / producer side).
        ``hl`` (lmi_gemm_skinny_hl, the decode precision mode): x holds 2 M rows [T(x); T(x - T(x))], 16-bit operand outputs are pairs again."""
        N, K = w.shape
        M = x.shape[0] // 2 if hl else x.shape[0]
        if packed is None:
            packed = getattr(w, "_lmi_packed", False)
        if hl:
            assert x.shape[0] == 2 * M and (epilogue in (1, 3) or out.shape[0] == 2 * M) and (norm_out is None or norm_out.shape[0] == 2 * M)
            self._check(self.lib.lmi_gemm_skinny_hl(_ptr(w), _ptr(x), _ptr(out), M, N, K, w.stride(0), x.stride(0), out.stride(0), int(epilogue),
                                                    int(bool(packed)), _ptr(rowsq_in), 0 if rowsq_in is None else rowsq_in.shape[1], int(norm_dim),
                                                    float(norm_eps), _ptr(norm_out), 0 if norm_out is None else norm_out.stride(0), _ptr(norm_gamma),
                                                    _ptr(rowsq_out), _DT[w.dtype], self._stream(out)))
            return out
        if rowsq_in is None and norm_out is None:
            self._check(self.lib.lmi_gemm_skinny(_ptr(w), _ptr(x), _ptr(out), M, N, K, w.stride(0), x.stride(0), out.stride(0), int(epilogue),
                                                 int(bool(packed)), _DT[w.dtype], self._stream(out)))
            return out
        self._check(self.lib.lmi_gemm_skinny_ex(_ptr(w), _ptr(x), _ptr(out), M, N, K, w.stride(0), x.stride(0), out.stride(0), int(epilogue),
                                                int(bool(packed)), _ptr(rowsq_in), 0 if rowsq_in is None else rowsq_in.shape[1], int(norm_dim),
                                                float(norm_eps), _ptr(norm_out), 0 if norm_out is None else norm_out.stride(0), _ptr(norm_gamma),
                                                _ptr(rowsq_out), _DT[w.dtype], self._stream(out)))
        return out

    def rope_qkv_skinny(self, w_rope, x, qkv, n_q_heads, n_kv_heads, head_dim, cos_all, sin_all, k_cache, v_cache, cache_stride, pos_rows, packed=None,
                        rowsq_in=None, norm_eps=0.0, hl=False):
        """lmi_rope_qkv_skinny: batched-decode q|k|v projection with RoPE + KV append in the epilogue (w_rope in rope_permute_rows order);
        rowsq_in: consumer side of the folded RMSNorm.  ``hl``: x holds 2 M rows [T(x); T(x - T(x))] (lmi_rope_qkv_skinny_hl); qkv has M rows."""
        M, K = (x.shape[0] // 2 if hl else x.shape[0]), w_rope.shape[1]
        if packed is None:
            packed = getattr(w_rope, "_lmi_packed", False)
        self._check((self.lib.lmi_rope_qkv_skinny_hl if hl else self.lib.lmi_rope_qkv_skinny)(_ptr(w_rope), _ptr(x), _ptr(qkv), M, n_q_heads, n_kv_heads, head_dim, K, w_rope.stride(0), x.stride(0),
                                                 qkv.stride(0), int(bool(packed)), _ptr(rowsq_in), 0 if rowsq_in is None else rowsq_in.shape[1],
                                                 float(norm_eps), _ptr(cos_all), _ptr(sin_all), _ptr(k_cache), _ptr(v_cache),
                                                 k_cache.stride(0), int(cache_stride), _ptr(pos_rows), _DT[w_rope.dtype], self._stream(qkv)))
        return qkv

    def llm_prefill_workspace(self, rows, hidden, n_q_heads, n_kv_heads, head_dim, ff, dtype):
        """lmi_llm_prefill_workspace_bytes -> (total bytes, [byte offset of LMI_WS_LLM_* buffer])."""
        import ctypes as C
        offs = (C.c_int64 * 6)()
        n = int(self.lib.lmi_llm_prefill_workspace_bytes(int(rows), hidden, n_q_heads, n_kv_heads, head_dim, ff, _DT[dtype], offs))
        if n < 0:
            raise RuntimeError("lmi_llm_prefill_workspace_bytes: " + self.lib.lmi_last_error().decode())
        return n, list(offs)

    def vit_workspace(self, rows, hidden, qkv_width, ff_padded, dtype):
        """lmi_vit_workspace_bytes -> (total bytes, [byte offset of LMI_WS_VIT_* buffer])."""
        import ctypes as C
        offs = (C.c_int64 * 4)()
        n = int(self.lib.lmi_vit_workspace_bytes(int(rows), hidden, qkv_width, ff_padded, _DT[dtype], offs))
        if n < 0:
            raise RuntimeError("lmi_vit_workspace_bytes: " + self.lib.lmi_last_error().decode())
        return n, list(offs)

    def decode_advance(self, logits, vocab, tok, pos, k_len=None, live=None, budget=None, eos=None, hist=None, hist_pos=None, suppress=None):
        """lmi_decode_advance: greedy choice + stop rule + position advance of B decode sequences, on the device (graph-capturable)."""
        B = logits.shape[0]
        self._check(self.lib.lmi_decode_advance(_ptr(logits), B, int(vocab), logits.stride(0), _ptr(suppress), 0 if suppress is None else suppress.numel(),
                                                _ptr(tok), _ptr(pos), _ptr(k_len), _ptr(live), _ptr(budget), _ptr(eos), 0 if eos is None else eos.numel(),
                                                _ptr(hist), _ptr(hist_pos), 0 if hist is None else hist.shape[0], self._stream(logits)))

    def decode_workspace_elems(self, q_rows, n_heads, head_dim, max_seqlen_k) -> int:
        n = int(self.lib.lmi_attn_decode_workspace_bytes(q_rows, n_heads, head_dim, max_seqlen_k))
        if n < 0:
            raise RuntimeError("lmi_attn_decode_workspace_bytes: bad arguments")
        return n // 4

    def rope_qk_at(self, qkv, n_q_heads, n_kv_heads, head_dim, cos_all, sin_all, k_cache, v_cache, pos_dev):
        """rope_qk with the position of row 0 taken from the device int32 ``pos_dev`` (graph-capturable decode step)."""
        self._check(self.lib.lmi_rope_qk_at(_ptr(qkv), qkv.shape[0], qkv.stride(0), n_q_heads, n_kv_heads, head_dim, _ptr(cos_all),
                                            _ptr(sin_all), _ptr(k_cache), _ptr(v_cache), k_cache.stride(0), _ptr(pos_dev),
                                            _DT[qkv.dtype], self._stream(qkv)))
        return qkv

    def rope_qk(self, qkv, n_q_heads, n_kv_heads, head_dim, cos, sin, k_cache=None, v_cache=None, cache_pos0=0):
        S = qkv.shape[0]
        ldc = 0 if k_cache is None else k_cache.stride(0)
        self._check(self.lib.lmi_rope_qk(_ptr(qkv), S, qkv.stride(0), n_q_heads, n_kv_heads, head_dim, _ptr(cos),
                                         _ptr(sin), _ptr(k_cache), _ptr(v_cache), ldc, cache_pos0, _DT[qkv.dtype],
                                         self._stream(qkv)))
        return qkv

    def embed_merge(self, ids, src, table, feats, out):
        S, D = out.shape
        ldf = 0 if feats is None else feats.stride(0)
        self._check(self.lib.lmi_embed_merge(_ptr(ids), _ptr(src), _ptr(table), _ptr(feats), _ptr(out), S, D, ldf,
                                             _DT[table.dtype], self._stream(out)))
        return out

    def gemv(self, w, x, out, bias=None, epilogue=0):
        N, K = w.shape
        assert not getattr(w, "_lmi_packed", False), "gemv: the GEMV kernels read the row-major layout"
        self._check(self.lib.lmi_gemv(_ptr(w), _ptr(x), _ptr_f32(bias), _ptr(out), N, K, w.stride(0), epilogue,
                                      _DT[w.dtype], self._stream(out)))
        return out

    def gemv_rmsnorm(self, w, x_f32, norm_weight, eps, out, epilogue=1):
        """out = epilogue(w @ rmsnorm(x_f32)): the decode step's norm + projection in one launch (K = 4096)."""
        N, K = w.shape
        assert not getattr(w, "_lmi_packed", False), "gemv_rmsnorm: the GEMV kernels read the row-major layout"
        self._check(self.lib.lmi_gemv_rmsnorm(_ptr(w), _ptr(x_f32), _ptr(norm_weight), float(eps), _ptr(out), N, K, w.stride(0),
                                              epilogue, _DT[w.dtype], self._stream(out)))
        return out

    def gemv_rmsnorm_rope(self, w_rope, x_f32, norm_weight, eps, qkv, n_q_heads, n_kv_heads, head_dim, cos_all, sin_all, k_cache, v_cache, pos_dev):
        """lmi_gemv_rmsnorm_rope: the decode step's q|k|v projection with the RMSNorm on its input and RoPE + KV append on its output."""
        assert not getattr(w_rope, "_lmi_packed", False), "gemv_rmsnorm_rope: the GEMV kernels read the row-major layout"
        self._check(self.lib.lmi_gemv_rmsnorm_rope(_ptr(w_rope), _ptr(x_f32), _ptr(norm_weight), float(eps), _ptr(qkv), n_q_heads, n_kv_heads,
                                                   head_dim, w_rope.shape[1], w_rope.stride(0), _ptr(cos_all), _ptr(sin_all), _ptr(k_cache),
                                                   _ptr(v_cache), k_cache.stride(0), _ptr(pos_dev), _DT[w_rope.dtype], self._stream(qkv)))
        return qkv

    def lm_head_last(self, w, x_f32, rows, norm_weight, eps, out):
        """out[r] = w @ rmsnorm(x_f32[rows[r]]) with the normalised row kept in fp32 (no activation rounding).
        x_f32: fp32 [S, K]; rows: int64 [n] on device or None (rows 0..n-1); out: fp32 [n, >= N]."""
        N, K = w.shape
        n = out.shape[0]
        assert out.dtype == torch.float32 and x_f32.dtype == torch.float32 and (rows is None or rows.dtype == torch.int64)
        assert not getattr(w, "_lmi_packed", False), "lm_head_last: the GEMV kernels read the row-major layout"
        self._check(self.lib.lmi_lm_head_last(_ptr(w), _ptr(x_f32), _ptr(rows), _ptr(norm_weight), float(eps), _ptr(out), n, N, K,
                                              w.stride(0), x_f32.stride(0), out.stride(0), _DT[w.dtype], self._stream(out)))
        return out

    def gemm_ex(self, a, w, out, bias=None, epilogue=EPI_STORE, act=ACT_NONE, rowsq_in=None, norm_dim=0, norm_eps=0.0, norm_out=None,
                norm_gamma=None, rowsq_out=None):
        """lmi_gemm with the RMSNorm folded in: consumer side (rowsq_in: [M, parts] fp32 partial sums of squares -> rows scaled
        by rstd) and / or producer side (RESIDUAL epilogue: norm_out = T(x * norm_gamma), rowsq_out [M, N/64])."""
        N, K = w.shape
        M = a.shape[0]
        parts = 0 if rowsq_in is None else rowsq_in.shape[1]
        if rowsq_in is not None:
            assert rowsq_in.dtype == torch.float32 and rowsq_in.is_contiguous() and rowsq_in.shape[0] >= M
        if norm_out is not None:
            assert rowsq_out is not None and rowsq_out.dtype == torch.float32 and rowsq_out.is_contiguous() and rowsq_out.shape == (M, N // 64)
        self._check(self.lib.lmi_gemm_ex(_ptr(a), _ptr(w), _ptr(out), _ptr_f32(bias), M, N, K, a.stride(0), self._ldw(w), out.stride(0), epilogue, act,
                                         _ptr(rowsq_in), parts, int(norm_dim), float(norm_eps), _ptr(norm_out), _ptr(norm_gamma), _ptr(rowsq_out),
                                         0 if norm_out is None else norm_out.stride(0), _DT[w.dtype], self._stream(out)))
        return out

    def rmsnorm_rope(self, a, w_qkv_rope, qkv, rowsq_in, eps, cos, sin, k_cache, v_cache, cache_pos0, n_q_heads, n_kv_heads, head_dim):
        """lmi_rmsnorm_rope: qkv = [RoPE(q), RoPE(k), v] of rstd * (a @ w^T), K / V appended to the cache rows cache_pos0..;
        w_qkv_rope in weights.rope_permute_rows order; rowsq_in None = a is already normalised."""
        M, K = a.shape[0], w_qkv_rope.shape[1]
        parts = 0 if rowsq_in is None else rowsq_in.shape[1]
        ldc = 0 if k_cache is None else k_cache.stride(0)
        assert qkv.shape[1] == (n_q_heads + 2 * n_kv_heads) * head_dim == w_qkv_rope.shape[0]
        self._check(self.lib.lmi_rmsnorm_rope(_ptr(a), _ptr(w_qkv_rope), _ptr(qkv), _ptr(rowsq_in), parts, float(eps), _ptr(cos), _ptr(sin),
                                              _ptr(k_cache), _ptr(v_cache), ldc, int(cache_pos0), M, n_q_heads, n_kv_heads, head_dim, K,
                                              a.stride(0), self._ldw(w_qkv_rope), qkv.stride(0), _DT[w_qkv_rope.dtype], self._stream(qkv)))
        return qkv

    def add_rmsnorm(self, x, delta, w, out, eps):
        """x (fp32, in place) += delta (compute type or fp32); out = rmsnorm(x) * w in the compute type (out None: add only)."""
        M, D = x.shape
        dt = _DT[out.dtype] if out is not None else (_DT[delta.dtype] if delta.dtype != torch.float32 else LMI_F16)
        self._check(self.lib.lmi_add_rmsnorm(_ptr(x), _ptr(delta), _DT[delta.dtype], _ptr(w), _ptr(out), M, D, x.stride(0), delta.stride(0),
                                             0 if out is None else out.stride(0), float(eps), dt, self._stream(x)))
        return out

    def quantize_fp8(self, x, out, scale: float):
        """out (uint8 view of fp8 e4m3fn, [M, D]) = fp8(x * scale); x fp32 / fp16 / bf16 [M, D]."""
        M, D = x.shape
        assert out.dtype in (torch.uint8, torch.float8_e4m3fn) and out.shape == x.shape
        self._check(self.lib.lmi_quantize_fp8(_ptr(x), _DT[x.dtype], _ptr(out), M, D, x.stride(0), out.stride(0), float(scale), self._stream(x)))
        return out

    def gemm_fp8(self, a8, w8, out, bias=None, epilogue=EPI_STORE, act=ACT_NONE, scale_exp: int = 0, out_scale: float = 1.0):
        """out = epilogue(2^scale_exp * (a8 @ w8.T)) on fp8 e4m3fn operands (uint8 / float8 tensors [M, K], [N, K]); a uint8 / float8
        ``out`` (STORE [+GELU-tanh], SWIGLU) receives fp8(result * out_scale)."""
        N, K = w8.shape
        M = a8.shape[0]
        if out.dtype in (torch.uint8, torch.float8_e4m3fn):
            dt = _lib.LMI_FP8
        else:
            dt = _DT[out.dtype] if out.dtype in (torch.float16, torch.bfloat16) else LMI_F16
        self._check(self.lib.lmi_gemm_fp8(_ptr(a8), _ptr(w8), _ptr(out), _ptr_f32(bias), M, N, K, a8.stride(0), w8.stride(0), out.stride(0),
                                          epilogue, act, int(scale_exp), dt, float(out_scale), self._stream(out)))
        return out

    # ---- low-bit correction phase (lmi_lo4) ------------------------------------------------------------------------------------------
    @staticmethod
    def _lo4_desc(a: Optional[Lo4Act], w4: Optional[Lo4Weight], out4: Optional[Lo4Act]) -> "_lib.Lo4Desc":
        d = _lib.Lo4Desc()
        if a is not None:
            d.a4, d.a4_scale, d.lda4, d.lds4, d.k4 = a.img.data_ptr(), a.sc.data_ptr(), a.img.stride(0), a.sc.stride(0), a.K4
        if w4 is not None:
            d.w4, d.w4_scale, d.ldw4 = w4.img.data_ptr(), w4.sc.data_ptr(), w4.img.stride(0)
        if out4 is not None:
            d.out4, d.out4_scale, d.ld_out4, d.ld_out4s = out4.img.data_ptr(), out4.sc.data_ptr(), out4.img.stride(0), out4.sc.stride(0)
        sel = a if (a is not None and a.row_sel is not None) else out4          # consumer and producer share ONE row space: the same selection
        if sel is not None and sel.row_sel is not None:
            assert out4 is None or a is None or out4.row_sel is None or out4.row_sel.data_ptr() == a.row_sel.data_ptr()
            d.row_sel, d.unit_sel = sel.row_sel.data_ptr(), sel.unit_sel.data_ptr()
            if sel.sel_ranges is not None and len(sel.sel_ranges):
                d._keep = sel.sel_ranges                              # host array, read during the call
                d.sel_ranges, d.n_sel_ranges = sel.sel_ranges.ctypes.data, len(sel.sel_ranges)
        return d

    def quantize_w4(self, w: torch.Tensor, head_pad: Optional[tuple] = None) -> Lo4Weight:
        """fp4 image + per-row E8M0 scales of a ROW-MAJOR 16-bit weight [N, K] (lmi_quantize_w4; once, at load).  ``head_pad`` = (n_heads,
        head_dim): the image in the per-head padded k order of an attention output's residual image (lmi_attn_varlen_fwd_lo4)."""
        assert not getattr(w, "_lmi_packed", False) and w.stride(1) == 1
        if head_pad is not None and head_pad[1] % 32:
            H, hd = head_pad
            nb = (hd + 31) // 32 * 32
            wp = torch.zeros(w.shape[0], lo4_head_k4(H, hd), dtype=w.dtype, device=w.device)
            wp[:, :H * nb].view(w.shape[0], H, nb)[:, :, :hd] = w.view(w.shape[0], H, hd)
            w = wp
        N, K = w.shape
        k4 = lo4_k4(K)
        img = torch.empty(N, k4 // 2, dtype=torch.uint8, device=w.device)
        sc = torch.empty(N, dtype=torch.uint8, device=w.device)
        self._check(self.lib.lmi_quantize_w4(_ptr(w), _ptr(img), _ptr(sc), N, K, k4, w.stride(0), img.stride(0), _DT[w.dtype], self._stream(w)))
        return Lo4Weight(img, sc)

    def attention_lo4(self, q, k, v, act: Lo4Act, cu_q, cu_k, max_seqlen_q, n_heads, n_kv_heads, head_dim, scale, causal, window=0):
        """lmi_attn_varlen_fwd_lo4: act.hi = the 16-bit attention output rows, act.img / act.sc = the image of their rounding residual in the
        per-head padded k order (act must have k4 = lo4_head_k4(n_heads, head_dim))."""
        n_seq = cu_q.numel() - 1
        assert act.K4 == lo4_head_k4(n_heads, head_dim)
        self._check(self.lib.lmi_attn_varlen_fwd_lo4_rows(_ptr(q), _ptr(k), _ptr(v), _ptr(act.hi), _ptr(act.img), _ptr(act.sc), act.img.stride(0),
                                                          act.sc.stride(0), _ptr(cu_q), _ptr(cu_k), n_seq, int(max_seqlen_q), n_heads, n_kv_heads, head_dim,
                                                          q.stride(0), k.stride(0), v.stride(0), act.hi.stride(0), float(scale), int(bool(causal)),
                                                          int(window), _ptr(act.row_sel), _DT[q.dtype], self._stream(act.hi)))
        return act

    def split_lo4(self, x_f32: torch.Tensor, act: Lo4Act) -> Lo4Act:
        """act = (T(x), fp4 image of x - T(x), block scales) of the fp32 x [M, K] (lmi_split_lo4)."""
        M, K = x_f32.shape
        assert x_f32.dtype == torch.float32 and act.hi.shape == (M, K)
        self._check(self.lib.lmi_split_lo4(_ptr(x_f32), _ptr(act.hi), _ptr(act.img), _ptr(act.sc), M, K, act.K4, x_f32.stride(0), act.hi.stride(0),
                                           act.img.stride(0), act.sc.stride(0), _DT[act.hi.dtype], self._stream(x_f32)))
        return act

    def norm_lo4(self, x, w, b, act: Lo4Act, eps) -> Lo4Act:
        """LayerNorm (b given) / RMSNorm (b None) of the fp32 rows x, handed over as a Lo4Act (lmi_norm_lo4)."""
        M, D = x.shape
        assert act.hi.shape == (M, D)
        self._check(self.lib.lmi_norm_lo4_rows(_ptr(x), _ptr(w), _ptr(b), _ptr(act.hi), _ptr(act.img), _ptr(act.sc), M, D, act.K4, x.stride(0),
                                               act.hi.stride(0), act.img.stride(0), act.sc.stride(0), float(eps), _ptr(act.row_sel), _DT[act.hi.dtype],
                                               self._stream(x)))
        return act

    def add_rmsnorm_lo4(self, x, delta, w, act: Lo4Act, eps) -> Lo4Act:
        """x (fp32, in place) += delta; act = the Lo4 pair of rmsnorm(x) * w (lmi_add_rmsnorm_lo4: the tensor-parallel layer's norm)."""
        M, D = x.shape
        assert act.hi.shape == (M, D)
        self._check(self.lib.lmi_add_rmsnorm_lo4(_ptr(x), _ptr(delta), _DT[delta.dtype], _ptr(w), _ptr(act.hi), _ptr(act.img), _ptr(act.sc), M, D, act.K4,
                                                 x.stride(0), delta.stride(0), act.hi.stride(0), act.img.stride(0), act.sc.stride(0), float(eps),
                                                 _DT[act.hi.dtype], self._stream(x)))
        return act

    def gemm_lo4(self, a: Lo4Act, w, w4: Lo4Weight, out, bias=None, epilogue=EPI_STORE, act=ACT_NONE, rowsq_in=None, norm_dim=0, norm_eps=0.0,
                 norm_out=None, norm_gamma=None, rowsq_out=None, out4: Optional[Lo4Act] = None):
        """lmi_gemm_lo4: gemm_ex on a.hi x w plus the correction phase a.img x w4.img into the same accumulators.  ``out4``: the Lo4Act whose
        ``hi`` is this launch's 16-bit result (``out`` for STORE / SWIGLU, ``norm_out`` for the RESIDUAL producer mode) receives the image of
        its residual."""
        N, K = w.shape
        M = a.hi.shape[0]
        parts = 0 if rowsq_in is None else rowsq_in.shape[1]
        d = self._lo4_desc(a, w4, out4)
        self._check(self.lib.lmi_gemm_lo4(_ptr(a.hi), _ptr(w), _ptr(out), _ptr_f32(bias), M, N, K, a.hi.stride(0), self._ldw(w), out.stride(0), epilogue, act,
                                          _ptr(rowsq_in), parts, int(norm_dim), float(norm_eps), _ptr(norm_out), _ptr(norm_gamma), _ptr(rowsq_out),
                                          0 if norm_out is None else norm_out.stride(0), C.byref(d), _DT[w.dtype], self._stream(out)))
        return out

    def rmsnorm_rope_lo4(self, a: Lo4Act, w_qkv_rope, w4: Lo4Weight, qkv, rowsq_in, eps, cos, sin, k_cache, v_cache, cache_pos0, n_q_heads, n_kv_heads,
                         head_dim):
        """lmi_rmsnorm_rope_lo4: rmsnorm_rope with the correction phase."""
        M, K = a.hi.shape[0], w_qkv_rope.shape[1]
        parts = 0 if rowsq_in is None else rowsq_in.shape[1]
        ldc = 0 if k_cache is None else k_cache.stride(0)
        d = self._lo4_desc(a, w4, None)
        self._check(self.lib.lmi_rmsnorm_rope_lo4(_ptr(a.hi), _ptr(w_qkv_rope), _ptr(qkv), _ptr(rowsq_in), parts, float(eps), _ptr(cos), _ptr(sin),
                                                  _ptr(k_cache), _ptr(v_cache), ldc, int(cache_pos0), M, n_q_heads, n_kv_heads, head_dim, K,
                                                  a.hi.stride(0), self._ldw(w_qkv_rope), qkv.stride(0), C.byref(d), _DT[w_qkv_rope.dtype],
                                                  self._stream(qkv)))
        return qkv

    def split_rows_hl(self, x_f32, out):
        """out [2 M, K] (16-bit) = rows T(x), then rows T(x - T(x)) of the fp32 x [M, K] (lmi_split_rows_hl; the decode precision mode)."""
        M, K = x_f32.shape
        assert x_f32.dtype == torch.float32 and out.shape == (2 * M, K)
        self._check(self.lib.lmi_split_rows_hl(_ptr(x_f32), _ptr(out), M, K, x_f32.stride(0), out.stride(0), _DT[out.dtype], self._stream(out)))
        return out

    def split_hi_lo(self, x_f32, out):
        """out [M, 2K] (16-bit) = [T(x) | T(x - T(x))] of the fp32 x [M, K] (lmi_split_hi_lo; split-operand precision mode)."""
        M, K = x_f32.shape
        assert x_f32.dtype == torch.float32 and out.shape == (M, 2 * K)
        self._check(self.lib.lmi_split_hi_lo(_ptr(x_f32), _ptr(out), M, K, x_f32.stride(0), out.stride(0), _DT[out.dtype], self._stream(out)))
        return out

    def attention_f32out(self, q, k, v, out32, cu_q, cu_k, max_seqlen_q, n_heads, n_kv_heads, head_dim, scale, causal, window=0):
        """lmi_attn_varlen_fwd_f32: the normalised attention output in fp32 [total_q, n_heads * head_dim]."""
        n_seq = cu_q.numel() - 1
        assert out32.dtype == torch.float32
        self._check(self.lib.lmi_attn_varlen_fwd_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(out32), out32.stride(0), _ptr(cu_q), _ptr(cu_k), n_seq,
                                                     int(max_seqlen_q), n_heads, n_kv_heads, head_dim, q.stride(0), k.stride(0), v.stride(0),
                                                     float(scale), int(bool(causal)), int(window), _DT[q.dtype], self._stream(out32)))
        return out32

    def attn_prep_fp8(self, qkv, cu, tile_base, n_tiles, n_q_heads, n_kv_heads, head_dim, q_scale, k_scale, v_scale, q8, k_img, v_img):
        """lmi_attn_prep_fp8: rotated q | k | v rows -> e4m3 q rows + per (kv head, 64-key tile) K / transposed-V LDS images (uint8 buffers)."""
        self._check(self.lib.lmi_attn_prep_fp8(_ptr(qkv), qkv.stride(0), _ptr(cu), _ptr(tile_base), cu.numel() - 1, int(n_tiles), n_q_heads, n_kv_heads,
                                               head_dim, float(q_scale), float(k_scale), float(v_scale), _ptr(q8), q8.stride(0), _ptr(k_img), _ptr(v_img),
                                               _DT[qkv.dtype], self._stream(q8)))

    def attention_fp8(self, q8, k_img, v_img, out, cu, tile_base, n_tiles, max_seqlen, n_heads, n_kv_heads, head_dim, scale, q_scale, k_scale, v_scale,
                      causal=True, out_fp8_scale=None, dtype=None):
        """lmi_attn_fp8_fwd: QK^T and PV on the fp8 matrix pipe from the operands of attn_prep_fp8.  out: T rows, or (out_fp8_scale given) uint8
        rows = e4m3(O * out_fp8_scale); dtype: the 16-bit type of the schedule when the output is fp8."""
        f8 = out_fp8_scale is not None
        self._check(self.lib.lmi_attn_fp8_fwd(_ptr(q8), q8.stride(0), _ptr(k_img), _ptr(v_img), None if f8 else _ptr(out), 0 if f8 else out.stride(0),
                                              _ptr(out) if f8 else None, out.stride(0) if f8 else 0, float(out_fp8_scale or 0.0), _ptr(cu), _ptr(tile_base),
                                              cu.numel() - 1, int(n_tiles), int(max_seqlen), n_heads, n_kv_heads, head_dim, float(scale), float(q_scale),
                                              float(k_scale), float(v_scale), int(bool(causal)), _DT[dtype if f8 else out.dtype], self._stream(out)))
        return out

    def attention_fp8out(self, q, k, v, out8, out_scale: float, cu_q, cu_k, max_seqlen_q, n_heads, n_kv_heads, head_dim, scale, causal, window=0):
        """lmi_attn_varlen_fwd_fp8: the attention output written as e4m3(O * out_scale) bytes (uint8 [total_q, n_heads * head_dim])."""
        n_seq = cu_q.numel() - 1
        self._check(self.lib.lmi_attn_varlen_fwd_fp8(_ptr(q), _ptr(k), _ptr(v), _ptr(out8), out8.stride(0), float(out_scale), _ptr(cu_q), _ptr(cu_k),
                                                     n_seq, int(max_seqlen_q), n_heads, n_kv_heads, head_dim, q.stride(0), k.stride(0), v.stride(0),
                                                     float(scale), int(bool(causal)), int(window), _DT[q.dtype], self._stream(out8)))
        return out8

    def rope_qkv_fp8(self, a8, w8_rope, qkv, scale_exp: int, cos, sin, k_cache, v_cache, cache_pos0, n_q_heads, n_kv_heads, head_dim):
        """lmi_rope_qkv_fp8: qkv = [RoPE(q), RoPE(k), v] of 2^scale_exp * (a8 @ w8_rope^T) on fp8 operands, K / V appended to the cache."""
        M, K = a8.shape[0], w8_rope.shape[1]
        ldc = 0 if k_cache is None else k_cache.stride(0)
        self._check(self.lib.lmi_rope_qkv_fp8(_ptr(a8), _ptr(w8_rope), _ptr(qkv), int(scale_exp), _ptr(cos), _ptr(sin), _ptr(k_cache), _ptr(v_cache), ldc,
                                              int(cache_pos0), M, n_q_heads, n_kv_heads, head_dim, K, a8.stride(0), w8_rope.stride(0), qkv.stride(0),
                                              _DT[qkv.dtype], self._stream(qkv)))
        return qkv

    def norm_fp8(self, x, w, b, out8, eps, out_scale: float):
        """out8 (uint8 / float8 [M, D]) = fp8(norm(x) * out_scale): LayerNorm when b is given, RMSNorm when b is None."""
        M, D = x.shape
        self._check(self.lib.lmi_norm_fp8(_ptr(x), _ptr(w), _ptr(b), _ptr(out8), M, D, x.stride(0), out8.stride(0), float(eps),
                                          float(out_scale), self._stream(out8)))
        return out8

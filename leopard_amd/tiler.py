"""Adaptive high-resolution multi-image tiler (host side, integer planning + PIL pixel work).

Mirrors, result-for-result, the published evaluation script of the reference
(evaluations/models/llava_multiimg_siglip_anyres.py, cited below as EVAL):

* ``plan_tile_budget``        <-> ``allocate_patches``           EVAL:26-58
* ``choose_canvas``           <-> ``select_best_resolution``     EVAL:61-99
* ``letterbox``               <-> ``resize_and_pad_image``       EVAL:102-140
* ``cut_tiles``               <-> ``divide_to_patches``          EVAL:143-162
* ``tile_sample``             <-> the per-sample block           EVAL:386-401
* ``siglip_preprocess``       <-> SiglipImageProcessor.preprocess call at EVAL:403-404

The quirks of the published code are part of the contract: PIL's (W, H) size tuple is
unpacked as (height, width) in the budget planner (harmless, the product is symmetric); Python's
banker's ``round``; an image whose natural tiling is exactly one tile gets no tiles at all (thumbnail
only); the 1x1 grid is never a candidate canvas; with >= 50 images no tiling happens.

Nothing here is on the GPU: the planner is integer arithmetic that takes microseconds, and the
resize uses PIL so that the u8 tiles are bit-identical to what the reference feeds its ViT.
"""
from __future__ import annotations

import functools
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

TILE = 364            # EVAL:26,61,394
SAMPLE_BUDGET = 50    # EVAL:387  (ViT inputs per sample, thumbnails included)


# --------------------------------------------------------------------------------------------------
# integer planning
# --------------------------------------------------------------------------------------------------
def plan_tile_budget(sizes: Sequence[Tuple[int, int]], tile: int = TILE, budget: int = SAMPLE_BUDGET) -> List[int]:
    """Per-image tile allowance.  ``sizes`` are PIL ``image.size`` tuples.  (EVAL:26-58)"""
    natural = []
    for a, b in sizes:
        n = round(a / tile) * round(b / tile)      # banker's rounding, as Python's round()
        natural.append(0 if n == 1 else n)
    total = sum(natural)
    if total <= budget:
        return natural
    ratio = budget / total
    scaled = [int(n * ratio) for n in natural]
    # The reference then trims one tile from each non-empty image in turn while the sum still
    # exceeds the budget.  floor() already guarantees sum <= budget, so this is a no-op kept only
    # to state the behaviour completely.
    while sum(scaled) > budget:
        over = sum(scaled) - budget
        for i, n in enumerate(scaled):
            if n > 0:
                scaled[i] = n - 1
                over -= 1
            if over == 0:
                break
    return scaled


def choose_canvas(size: Tuple[int, int], n_tiles: int, tile: int = TILE) -> Optional[Tuple[int, int]]:
    """Canvas (W, H), a multiple of ``tile`` with rows*cols <= n_tiles, that keeps the most source
    pixels after an aspect-preserving downscale; ties go to the least padded canvas; the first
    candidate in (rows outer, cols inner) order wins exact ties.  1x1 is never considered.  (EVAL:61-99)"""
    if n_tiles == 0:
        return None
    w0, h0 = size
    best, best_eff, best_waste = None, 0, float("inf")
    for rows in range(1, n_tiles + 1):
        for cols in range(1, n_tiles // rows + 1):
            if rows == 1 and cols == 1:
                continue
            cw, ch = cols * tile, rows * tile
            s = min(cw / w0, ch / h0)
            eff = min(int(w0 * s) * int(h0 * s), w0 * h0)
            waste = cw * ch - eff
            if eff > best_eff or (eff == best_eff and waste < best_waste):
                best, best_eff, best_waste = (cw, ch), eff, waste
    return best


@dataclass
class TilePlan:
    """Tiling decision for one sample (all integers)."""
    sizes: List[Tuple[int, int]]
    allowance: List[int]
    canvases: List[Optional[Tuple[int, int]]]
    tiles_per_image: List[int]          # real tile count (cols*rows of the canvas), thumbnail excluded

    @property
    def vit_inputs_per_image(self) -> List[int]:
        return [1 + t for t in self.tiles_per_image]

    @property
    def n_vit_inputs(self) -> int:
        return sum(self.vit_inputs_per_image)


def plan_sample(sizes: Sequence[Tuple[int, int]], tile: int = TILE, sample_budget: int = SAMPLE_BUDGET) -> TilePlan:
    """The integer part of EVAL:386-401 for one sample."""
    sizes = [tuple(s) for s in sizes]
    budget = sample_budget - len(sizes)
    if budget <= 0:
        # EVAL:400-401 — no tiling; the reference then claims one tile per image in the prompt
        # (which its own merge rejects); we report the pixel truth: thumbnails only.
        return TilePlan(sizes, [0] * len(sizes), [None] * len(sizes), [0] * len(sizes))
    allowance = plan_tile_budget(sizes, tile, budget)
    canvases = [choose_canvas(s, n, tile) for s, n in zip(sizes, allowance)]
    tiles = [0 if c is None else (c[0] // tile) * (c[1] // tile) for c in canvases]
    return TilePlan(sizes, allowance, canvases, tiles)


# --------------------------------------------------------------------------------------------------
# pixel work (PIL, u8)
# --------------------------------------------------------------------------------------------------
def letterbox(image, canvas: Optional[Tuple[int, int]]):
    """Aspect-preserving bicubic resize (PIL default filter for ``Image.resize``) so that one side
    fills the canvas, centred on black.  (EVAL:102-140)"""
    if canvas is None:
        return None
    from PIL import Image
    nw, nh, px, py = letterbox_geometry(image.size, canvas)
    out = Image.new("RGB", canvas, (0, 0, 0))
    out.paste(image.resize((nw, nh)), (px, py))
    return out


def cut_tiles(image, tile: int = TILE) -> list:
    """Row-major tile crops.  (EVAL:143-162)"""
    w, h = image.size
    return [image.crop((x, y, x + tile, y + tile)) for y in range(0, h, tile) for x in range(0, w, tile)]


def tile_sample(images: Sequence, tile: int = TILE, sample_budget: int = SAMPLE_BUDGET):
    """Returns (vit_inputs, plan): per image ``[original] + tiles`` in order (EVAL:396-399)."""
    plan = plan_sample([im.size for im in images], tile, sample_budget)
    vit_inputs = []
    for im, canvas in zip(images, plan.canvases):
        vit_inputs.append(im)
        if canvas is not None:
            vit_inputs.extend(cut_tiles(letterbox(im, canvas), tile))
    return vit_inputs, plan


def to_u8_tiles(vit_inputs: Sequence, size: int = TILE) -> np.ndarray:
    """Resize every ViT input to ``size x size`` with PIL bicubic (identity for tiles, aspect-squashing
    for the thumbnail — what SiglipImageProcessor(do_resize, resample=BICUBIC) does, EVAL:403-404) and
    stack as ``[N, size, size, 3]`` u8.  This is the buffer handed to the GPU."""
    from PIL import Image
    out = np.empty((len(vit_inputs), size, size, 3), dtype=np.uint8)
    for i, im in enumerate(vit_inputs):
        if im.size != (size, size):
            im = im.resize((size, size), resample=Image.BICUBIC)
        out[i] = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return out


def siglip_normalize(u8_tiles: np.ndarray) -> np.ndarray:
    """``(x/255 - 0.5) / 0.5`` and HWC->CHW in fp32: the host statement of the processor's
    rescale+normalize.  The GPU path does this inside ``lmi_preprocess_tiles``."""
    x = u8_tiles.astype(np.float32) * np.float32(1.0 / 255.0)
    x = (x - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def siglip_preprocess(vit_inputs: Sequence, size: int = TILE) -> np.ndarray:
    """[N,3,size,size] fp32 pixel_values, the tensor the reference passes to ``generate``."""
    return siglip_normalize(to_u8_tiles(vit_inputs, size))


# --------------------------------------------------------------------------------------------------
# Pillow's resampling taps (for the GPU tiler: leopard_amd/gpu_tiler.py + lmi_resample_u8)
# --------------------------------------------------------------------------------------------------
_PRECISION_BITS = 32 - 8 - 2          # libImaging Resample.c: taps are 22-bit fixed point for 8-bit channels


def _bicubic(x: np.ndarray) -> np.ndarray:
    """libImaging bicubic_filter (a = -0.5), same operation order in float64."""
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


@functools.lru_cache(maxsize=256)
def pil_resample_coeffs(in_size: int, out_size: int):
    """(bounds int32 [out, 2], taps int32 [out, ksize]) of ``Image.resize`` with BICUBIC along one axis from ``in_size``
    to ``out_size`` samples: a restatement of libImaging's precompute_coeffs + normalize_coeffs_8bpc (Pillow
    src/libImaging/Resample.c; third-party, absent from /root/reference — pinned by comparing the GPU tiler's output with
    PIL's own, bit for bit, in tests/test_gpu_tiler.py).  The arithmetic is float64 in the same order as the C code."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale                              # bicubic support = 2
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = (center - support + 0.5).astype(np.int64)        # C (int) cast: truncation toward zero
    xmin = np.maximum(xmin, 0)
    xmax = (center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    k = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):                                   # sequential accumulation order of the C loop
        w = _bicubic((x + xmin - center + 0.5) * ss)
        w = np.where(x < xmax, w, 0.0)
        k[:, x] = w
        ww = ww + w
    nz = ww != 0.0
    k[nz] = k[nz] / ww[nz, None]
    fixed = np.where(k < 0, -0.5 + k * (1 << _PRECISION_BITS), 0.5 + k * (1 << _PRECISION_BITS)).astype(np.int64)   # (int) truncation
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, fixed.astype(np.int32)


def letterbox_geometry(size: Tuple[int, int], canvas: Tuple[int, int]) -> Tuple[int, int, int, int]:
    """(new_w, new_h, paste_x, paste_y) of resize_and_pad_image for an image of PIL size ``size`` (EVAL:102-140)."""
    w0, h0 = size
    cw, ch = canvas
    sw, sh = cw / w0, ch / h0
    if sw < sh:
        nw, nh = cw, min(math.ceil(h0 * sw), ch)
    else:
        nw, nh = min(math.ceil(w0 * sh), cw), ch
    return nw, nh, (cw - nw) // 2, (ch - nh) // 2

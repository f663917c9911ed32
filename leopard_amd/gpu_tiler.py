"""The tiler on the GPU: resize + pad + crop of every image of a sample on the device, from the raw
u8 pixels to the ``[N, 364, 364, 3]`` u8 tile stack that ``lmi_preprocess_tiles`` normalises and patchifies.

Same results as the host path ``tiler.tile_sample`` + ``tiler.to_u8_tiles`` (which use PIL), bit for bit: the integer plan
comes from the same planner, the taps are Pillow's (``tiler.pil_resample_coeffs``), and ``lmi_resample_u8`` is
libImaging's fixed-point arithmetic.  Replaces, per sample, N PIL resizes on the host (about 95 ms at C3) and the 16.7 MB
tile upload by one upload of the source pixels and a few HBM-bound kernels.

Reference: resize_and_pad_image / divide_to_patches / the per-sample block and the SiglipImageProcessor resize,
evaluations/models/llava_multiimg_siglip_anyres.py:102-162, 386-405.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import tiler
from .ops import Ops


class GpuTiler:
    def __init__(self, ops: Ops, device, tile: int = tiler.TILE, sample_budget: int = tiler.SAMPLE_BUDGET, out_size: int = 0):
        """``out_size``: side of the ViT inputs when it differs from the planner's tile (the processor then resizes every
        input, tiles included — never the case for Leopard's 364 / 364, used by reduced test configurations)."""
        self.ops, self.device, self.tile, self.sample_budget = ops, torch.device(device), tile, sample_budget
        self.out_size = out_size or tile
        self._coef = {}

    def _coeffs(self, in_size: int, out_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        key = (in_size, out_size)
        if key not in self._coef:
            bounds, taps = tiler.pil_resample_coeffs(in_size, out_size)
            self._coef[key] = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(np.ascontiguousarray(taps)).to(self.device))
        return self._coef[key]

    def _resize_into(self, img: torch.Tensor, dst: torch.Tensor):
        """PIL ``img.resize((dst_w, dst_h))`` (BICUBIC) written into the u8 window ``dst`` [dst_h, dst_w, 3]: the row pass
        first, then the column pass, each skipped when that axis keeps its size (as ImagingResample does)."""
        h, w = img.shape[0], img.shape[1]
        dh, dw = dst.shape[0], dst.shape[1]
        if (dh, dw) == (h, w):
            dst.copy_(img)
            return
        cur = img
        if dw != w:
            tgt = dst if dh == h else torch.empty((h, dw, 3), dtype=torch.uint8, device=self.device)
            self.ops.resample_u8(cur, tgt, 0, *self._coeffs(w, dw))
            cur = tgt
        if dh != h:
            self.ops.resample_u8(cur, dst, 1, *self._coeffs(h, dh))

    def tile_sample(self, images: Sequence[np.ndarray]):
        """images: u8 HWC arrays (RGB) on the host, or u8 HWC tensors already resident on the device.  Returns
        (tiles u8 [N, tile, tile, 3] on the device, plan) in the reference's ViT input order: per image the squashed whole
        image, then its tiles row-major."""
        T = self.tile
        sizes = [(int(im.shape[1]), int(im.shape[0])) for im in images]            # PIL (W, H)
        plan = tiler.plan_sample(sizes, T, self.sample_budget)
        tiles = torch.empty((plan.n_vit_inputs, self.out_size, self.out_size, 3), dtype=torch.uint8, device=self.device)
        n = 0
        for im, size, canvas in zip(images, sizes, plan.canvases):
            if torch.is_tensor(im):
                src = im if im.device == self.device else im.to(self.device, non_blocking=True)
            else:
                src = torch.from_numpy(np.array(im, dtype=np.uint8, order="C"))         # a writable copy (images may be read-only views)
                src = src.to(self.device, non_blocking=True) if self.device.type != "cpu" else src
            self._resize_into(src, tiles[n])                                       # the thumbnail: aspect-squashing resize
            n += 1
            if canvas is None:
                continue
            cw, ch = canvas
            nw, nh, px, py = tiler.letterbox_geometry(size, canvas)
            board = torch.zeros((ch, cw, 3), dtype=torch.uint8, device=self.device)    # black canvas
            self._resize_into(src, board[py:py + nh, px:px + nw])
            for y in range(0, ch, T):
                for x in range(0, cw, T):
                    self._resize_into(board[y:y + T, x:x + T], tiles[n])              # a copy when the sizes agree
                    n += 1
        assert n == plan.n_vit_inputs
        return tiles, plan

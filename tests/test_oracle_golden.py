"""Pin the CPU oracle (oracle/leopard_oracle.py) against fixtures generated FROM THE REFERENCE
(oracle/gen_golden.py): integers bit-exact, fp32 tensors <= 1e-5."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from leopard_amd.config import full_config, tiny_config
from leopard_amd.synth import (KIND_BIAS, KIND_WEIGHT, param_specs, synth_array, synth_image_u8,
                               synth_state_dict_numpy)
from oracle import leopard_oracle as O

FP32_TOL = 1e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_tiler_plans_bit_exact(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "tiler_plans.json")))
    for row in g["plans"]:
        sizes = [tuple(s) for s in row["sizes"]]
        alloc = O.allocate_patches(sizes, patch_budget=row["budget"])
        assert alloc == row["allocate"]
        res = [O.select_best_resolution(s, n) for s, n in zip(sizes, alloc)]
        assert [None if r is None else list(r) for r in res] == row["resolution"]
    for row in g["resolution_sweep"]:
        r = O.select_best_resolution(tuple(row["size"]), row["n"])
        assert (None if r is None else list(r)) == row["resolution"]
    for row in g["tight"]:
        assert O.allocate_patches([tuple(s) for s in row["sizes"]], patch_budget=row["budget"]) == row["allocate"]


def test_known_answer_plans():
    """SURVEY.md Appendix B table (computed with the reference functions)."""
    assert O.allocate_patches([(336, 336)], patch_budget=49) == [0]
    assert O.allocate_patches([(1344, 896)], patch_budget=49) == [8]
    assert O.allocate_patches([(1344, 896)] * 6, patch_budget=44) == [7] * 6
    assert O.allocate_patches([(1344, 896)] * 8, patch_budget=42) == [5] * 8
    assert O.allocate_patches([(1344, 896)] * 20, patch_budget=30) == [1] * 20
    for n in (6, 7, 8):
        assert O.select_best_resolution((1344, 896), n) == (1092, 728)
    assert O.select_best_resolution((1344, 896), 5) == (728, 728)
    assert O.select_best_resolution((1344, 896), 12) == (1456, 1092)
    assert O.select_best_resolution((1344, 896), 1) is None


def test_tile_pixels_bit_exact(golden_dir):
    from PIL import Image
    meta = json.load(open(os.path.join(golden_dir, "tiles_meta.json")))
    full = _load(golden_dir, "tiles_seed2.npz")["tiles_seed2"]
    for m in meta:
        im = Image.fromarray(synth_image_u8(m["seed"], m["w"], m["h"]))
        res = O.select_best_resolution(im.size, m["n"])
        assert (None if res is None else list(res)) == m["resolution"]
        tiles = O.divide_to_patches(O.resize_and_pad_image(im, res), 364)
        shas = [hashlib.sha256(np.asarray(t, dtype=np.uint8).tobytes()).hexdigest() for t in tiles]
        assert shas == m["tile_sha256"]
        if m["seed"] == 2:
            assert np.array_equal(np.stack([np.asarray(t) for t in tiles]), full)


def test_image_processor(golden_dir):
    from PIL import Image
    g = _load(golden_dir, "image_processor.npz")
    ims = [Image.fromarray(synth_image_u8(7, 1344, 896)), Image.fromarray(synth_image_u8(8, 364, 364)),
           Image.fromarray(synth_image_u8(9, 336, 336))]
    outs = [O.siglip_image_processor(im)[0].numpy() for im in ims]
    assert list(outs[0].shape) == list(g["shape"])
    assert np.abs(outs[0][:, :48, :48] - g["crop0"]).max() <= FP32_TOL
    assert np.abs(outs[1][:, 100:148, 200:248] - g["crop1"]).max() <= FP32_TOL
    assert np.abs(outs[2][:, -48:, -48:] - g["crop2"]).max() <= FP32_TOL


def test_pixel_shuffle_and_projector(golden_dir):
    g = _load(golden_dir, "pixel_shuffle_projector.npz")
    assert np.array_equal(O.pixel_shuffle(torch.from_numpy(g["ps_in1"])).numpy(), g["ps_out1"])
    x2 = torch.arange(676 * 4, dtype=torch.float32).reshape(1, 676, 4)
    assert np.array_equal(O.pixel_shuffle(x2).numpy(), g["ps_out2"])
    W = {"multi_modal_projector.linear_1.weight": synth_array("multi_modal_projector.linear_1.weight", (96, 256), KIND_WEIGHT),
         "multi_modal_projector.linear_1.bias": synth_array("multi_modal_projector.linear_1.bias", (96,), KIND_BIAS),
         "multi_modal_projector.linear_2.weight": synth_array("multi_modal_projector.linear_2.weight", (96, 96), KIND_WEIGHT),
         "multi_modal_projector.linear_2.bias": synth_array("multi_modal_projector.linear_2.bias", (96,), KIND_BIAS)}
    out = O.projector(torch.from_numpy(g["proj_in"]), O.weights_from_numpy(W))
    assert np.abs(out.numpy() - g["proj_out"]).max() <= FP32_TOL


def test_tiny_end_to_end_vs_reference_forward(golden_dir):
    """The reference forward (EVAL:201-361), run unmodified over third-party modules, vs the oracle."""
    g = _load(golden_dir, "tiny_e2e.npz")
    cfg = tiny_config()
    W = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    for name in "abc":
        ids = torch.from_numpy(g[f"{name}_ids"]).reshape(1, -1)
        pix = torch.from_numpy(g[f"{name}_pix"])
        logits, parts = O.prefill_logits(ids, pix, W, cfg, return_parts=True)
        assert np.abs(parts["vit"].numpy() - g[f"{name}_vit"]).max() <= FP32_TOL
        assert np.abs(parts["visual_tokens"].numpy() - g[f"{name}_vis"]).max() <= FP32_TOL
        assert np.abs(parts["inputs_embeds"].numpy() - g[f"{name}_embeds"]).max() <= FP32_TOL
        assert np.array_equal(parts["position_ids"].numpy(), g[f"{name}_pos"])
        assert g[f"{name}_mask"].min() == 1
        assert logits.shape == g[f"{name}_logits"].shape
        assert np.abs(logits.numpy() - g[f"{name}_logits"]).max() <= FP32_TOL
        last = O.prefill_logits(ids, pix, W, cfg, last_only=True)
        assert np.abs(last.numpy()[0, 0] - g[f"{name}_logits"][0, -1]).max() <= FP32_TOL
        gen = O.greedy_generate(ids, pix, W, cfg, max_new_tokens=4)
        assert np.array_equal(gen.numpy().reshape(-1), g[f"{name}_greedy"])


def test_merge_mismatch_raises(golden_dir):
    g = _load(golden_dir, "tiny_e2e.npz")
    assert int(g["mismatch_raises"]) == 1
    with pytest.raises(ValueError):
        O.merge_plan(np.array([1, 500, 2]), 500, 2 * 4, 4)


def test_fullwidth_layers_vs_third_party(golden_dir):
    """One full-width SigLIP layer and one full-width Llama layer (llama3 RoPE at positions 3000+)."""
    g = _load(golden_dir, "fullwidth_layers.npz")
    cfg = full_config()
    specs = {n: (s, k) for n, s, k in param_specs(cfg)}
    pre = "vision_tower.vision_model.encoder.layers.0."
    W = O.weights_from_numpy({k: synth_array(k, *specs[k]) for k in specs if k.startswith(pre)})
    x = torch.from_numpy(np.random.default_rng(21).standard_normal((2, 40, 1152)).astype(np.float32))
    with torch.no_grad():
        y = O.siglip_layer(x, W, 0, cfg)
    assert np.abs(y.numpy() - g["siglip_out"]).max() <= 2e-5
    pre = "language_model.model.layers.0."
    W = O.weights_from_numpy({k: synth_array(k, *specs[k]) for k in specs if k.startswith(pre)})
    S = 48
    xs = torch.from_numpy(np.random.default_rng(22).standard_normal((1, S, 4096)).astype(np.float32))
    pos = torch.arange(3000, 3000 + S).unsqueeze(0)
    tc = cfg.text_config
    inv = O.llama3_inv_freq(tc.head_dim, tc.rope_theta, tc.rope_scaling)
    assert np.abs(inv.numpy() - g["inv_freq"]).max() <= 1e-9
    cos, sin = O.rope_tables(pos, tc.head_dim, tc.rope_theta, tc.rope_scaling)
    with torch.no_grad():
        ys = O.llama_layer(xs, W, 0, cfg, cos, sin)
    assert np.abs(ys.numpy() - g["llama_out"]).max() <= 5e-5

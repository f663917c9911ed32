"""Leopard-Idefics2 engine over the emulated kernels vs the Idefics2 CPU oracle, at a micro configuration that satisfies the
kernel shape rules (vision 1152 = 16 x 72, perceiver head_dim 96, Mistral head_dim 128): two images of different sizes."""
import numpy as np
import pytest
import torch

from leopard_amd.config import Idefics2Config, PerceiverConfig, TextConfig, VisionConfig
from leopard_amd.idefics2 import (Idefics2Engine, Idefics2SynthSource, Idefics2Weights, navit_position_ids,
                                  preprocess_image_u8, resize_output_size)
from leopard_amd.synth import idefics2_state_dict_numpy, synth_image_u8
from oracle import idefics2_oracle as IO
from tests.emu_util import emu_ops


def micro_idefics2():
    return Idefics2Config(
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=1, num_attention_heads=16,
                                   image_size=56, patch_size=14),
        text_config=TextConfig(hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                               num_key_value_heads=1, vocab_size=256, rope_theta=10000.0, rope_scaling=None, sliding_window=6),
        perceiver_config=PerceiverConfig(n_latents=3, depth=2, n_heads=1, head_dim=96, num_key_value_heads=1),
        image_token_id=250, longest_edge=56)


def test_position_ids_and_size_rule_match_oracle():
    for nh, nw, g in [(46, 70, 70), (70, 46, 70), (3, 4, 4), (1, 1, 4), (70, 70, 70), (33, 17, 70)]:
        assert np.array_equal(navit_position_ids(nh, nw, g), IO.navit_position_ids(nh, nw, g).numpy())
    for h, w in [(896, 1344), (1344, 896), (300, 400), (5, 981), (3000, 3000)]:
        assert resize_output_size(h, w, 980) == IO.resize_output_size(h, w, 980)


def test_processor_u8_matches_oracle():
    from PIL import Image
    im = Image.fromarray(synth_image_u8(3, 1344, 896))
    u8 = preprocess_image_u8(im, 980)
    assert u8.shape == (653, 980, 3)
    ref = IO.image_processor(im, 980)
    got = (torch.from_numpy(u8.copy()).float() * (1.0 / 255.0) - 0.5) / 0.5
    assert torch.equal(got.permute(2, 0, 1), ref)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 3e-3), (torch.bfloat16, 2.4e-2)])
def test_idefics2_prefill_matches_oracle(dtype, tol):
    ops = emu_ops()
    cfg = micro_idefics2()
    W = Idefics2Weights.build(cfg, Idefics2SynthSource(cfg, ops, "cpu", dtype), dtype)
    eng = Idefics2Engine(cfg, W, ops=ops, device="cpu")
    rng = np.random.default_rng(8)
    img_a = torch.from_numpy(rng.standard_normal((3, 42, 56)).astype(np.float32))      # 3 x 4 patches
    img_b = torch.from_numpy(rng.standard_normal((3, 58, 30)).astype(np.float32))      # 4 x 2 patches + remainder pixels
    L = cfg.perceiver_config.n_latents
    ids = torch.tensor([[5, 7] + [250] * L + [9, 11, 13] + [250] * L + [17, 19]])      # 13 tokens > window 6
    res = eng.prefill(ids, [img_a, img_b], all_logits=True, keep_parts=True)
    Wt = IO.weights_from_numpy(idefics2_state_dict_numpy(cfg))
    logits, parts = IO.prefill_logits(ids, [img_a, img_b], Wt, cfg, return_parts=True)
    assert res.n_tiles == 2 and res.seq_len == ids.shape[1]
    assert (res.parts["image_features"] - parts["image_features"]).abs().max() <= tol * 2
    assert (res.logits_all - logits[0]).abs().max() <= tol
    with pytest.raises(ValueError, match="number of image tokens"):
        eng.prefill(ids, [img_a])

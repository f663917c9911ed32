"""Pin oracle/idefics2_oracle.py against the third-party Idefics2ForConditionalGeneration fixture (oracle/gen_golden.py):
two images of different sizes, processed padded+masked by the third-party model and unpadded by the restatement."""
import json
import os

import numpy as np
import torch

from leopard_amd.config import idefics2_tiny_config
from leopard_amd.synth import idefics2_state_dict_numpy
from oracle import idefics2_oracle as IO


def test_tiny_logits_and_image_features(golden_dir):
    g = np.load(os.path.join(golden_dir, "idefics2_tiny.npz"))
    cfg = idefics2_tiny_config()
    W = IO.weights_from_numpy(idefics2_state_dict_numpy(cfg))
    ids = torch.from_numpy(g["ids"]).reshape(1, -1)
    images = [torch.from_numpy(g["img_a"]), torch.from_numpy(g["img_b"])]
    logits, parts = IO.prefill_logits(ids, images, W, cfg, return_parts=True)
    assert np.abs(parts["image_features"].numpy() - g["image_hidden_states"]).max() <= 1e-5
    assert logits.shape == g["logits"].shape and np.abs(logits.numpy() - g["logits"]).max() <= 1e-5


def test_resize_rule(golden_dir):
    for w, h, ow, oh in json.load(open(os.path.join(golden_dir, "idefics2_resize.json"))):
        assert IO.resize_output_size(h, w, 980) == (oh, ow)
    assert IO.resize_output_size(896, 1344, 980) == (653, 980)          # 70 x 46 = 3220 patches
    assert (653 // 14) * (980 // 14) == 3220


def test_navit_position_ids():
    # full-resolution square image uses the identity grid; a half-height image skips every other row bucket
    assert IO.navit_position_ids(4, 4, 4).tolist() == list(range(16))
    assert IO.navit_position_ids(2, 4, 4).tolist() == [0, 1, 2, 3, 8, 9, 10, 11]
    ids = IO.navit_position_ids(46, 70, 70)
    assert ids.numel() == 3220 and int(ids.max()) < 4900 and ids[0] == 0


def test_patch_validity_any_rule_matches_third_party(golden_dir):
    """The 4.4x rule (a patch belongs to an image when ANY of its pixels is real): the smaller image of a mixed-size sample keeps a
    partly zero-padded patch column; fixture generated from the third-party model (oracle/gen_golden.py: the 5.15 model fed the mask
    that makes its ALL rule coincide with ANY)."""
    from leopard_amd.idefics2_compat import Idefics2ForConditionalGeneration as M
    g = np.load(os.path.join(golden_dir, "idefics2_tiny_any.npz"))
    cfg = idefics2_tiny_config()
    W = IO.weights_from_numpy(idefics2_state_dict_numpy(cfg))
    pix, msk = torch.from_numpy(g["pixel_values"]), torch.from_numpy(g["pixel_attention_mask"])
    ids = torch.from_numpy(g["ids"]).reshape(1, -1)
    any_imgs = M.unpad_images(pix, msk, "any", cfg.vision_config.patch_size)
    all_imgs = M.unpad_images(pix, msk, "all", cfg.vision_config.patch_size)
    assert [tuple(i.shape) for i in all_imgs] == [(3, 42, 56), (3, 58, 30)] and [tuple(i.shape) for i in any_imgs] == [(3, 42, 56), (3, 56, 42)]
    logits, parts = IO.prefill_logits(ids, any_imgs, W, cfg, return_parts=True)
    assert np.abs(parts["image_features"].numpy() - g["image_hidden_states"]).max() <= 1e-5
    assert np.abs(logits.numpy() - g["logits"]).max() <= 1e-5
    other = np.load(os.path.join(golden_dir, "idefics2_tiny.npz"))
    assert np.abs(g["logits"] - other["logits"]).max() > 1e-3                 # the two rules really differ on this sample

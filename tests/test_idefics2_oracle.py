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

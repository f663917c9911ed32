"""Host logic of the path — the product tiler (leopard_amd/tiler.py) and the harness counterpart
(leopard_amd/harness.py) — against fixtures captured FROM THE REFERENCE (oracle/gen_golden.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

from leopard_amd import harness as H
from leopard_amd import tiler as T
from leopard_amd.synth import synth_image_u8


def test_plans_bit_exact(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "tiler_plans.json")))
    for row in g["plans"]:
        sizes = [tuple(s) for s in row["sizes"]]
        plan = T.plan_sample(sizes)
        assert plan.allowance == row["allocate"]
        assert [None if c is None else list(c) for c in plan.canvases] == row["resolution"]
        assert plan.tiles_per_image == row["tiles"]
    for row in g["resolution_sweep"]:
        c = T.choose_canvas(tuple(row["size"]), row["n"])
        assert (None if c is None else list(c)) == row["resolution"]
    for row in g["tight"]:
        assert T.plan_tile_budget([tuple(s) for s in row["sizes"]], budget=row["budget"]) == row["allocate"]


def test_baseline_config_plans():
    """SURVEY.md Appendix B."""
    for n_img, (n_vit, per) in {1: (7, 6), 4: (28, 6), 6: (42, 6), 8: (40, 4), 20: (20, 0), 49: (49, 0)}.items():
        p = T.plan_sample([(1344, 896)] * n_img)
        assert p.n_vit_inputs == n_vit and p.tiles_per_image == [per] * n_img
    assert T.plan_sample([(336, 336)]).n_vit_inputs == 1
    assert T.plan_sample([(100, 100)] * 50).tiles_per_image == [0] * 50          # budget <= 0: no tiling


def test_tile_pixels_bit_exact(golden_dir):
    from PIL import Image
    meta = json.load(open(os.path.join(golden_dir, "tiles_meta.json")))
    full = np.load(os.path.join(golden_dir, "tiles_seed2.npz"))["tiles_seed2"]
    for m in meta:
        im = Image.fromarray(synth_image_u8(m["seed"], m["w"], m["h"]))
        canvas = T.choose_canvas(im.size, m["n"])
        tiles = T.cut_tiles(T.letterbox(im, canvas))
        assert [hashlib.sha256(np.asarray(t, dtype=np.uint8).tobytes()).hexdigest() for t in tiles] == m["tile_sha256"]
        if m["seed"] == 2:
            assert np.array_equal(T.to_u8_tiles(tiles), full)


def test_preprocess_matches_third_party_processor(golden_dir):
    from PIL import Image
    g = np.load(os.path.join(golden_dir, "image_processor.npz"))
    ims = [Image.fromarray(synth_image_u8(7, 1344, 896)), Image.fromarray(synth_image_u8(8, 364, 364)),
           Image.fromarray(synth_image_u8(9, 336, 336))]
    out = T.siglip_preprocess(ims)
    assert list(out.shape[1:]) == list(g["shape"])
    assert np.abs(out[0][:, :48, :48] - g["crop0"]).max() <= 1e-6
    assert np.abs(out[1][:, 100:148, 200:248] - g["crop1"]).max() <= 1e-6
    assert np.abs(out[2][:, -48:, -48:] - g["crop2"]).max() <= 1e-6


def test_harness_matches_reference_capture(golden_dir):
    """Prompt string, ViT-input order/sizes, image-token count, generate kwargs, result-row schema and file name,
    for 20 synthetic records x 3 settings — including the records the reference itself crashes on."""
    from PIL import Image
    cap = json.load(open(os.path.join(golden_dir, "harness_capture.json")))
    for setting in cap["settings"]:
        for rec, got in zip(cap["records"], setting["per_record"]):
            images = [Image.fromarray(synth_image_u8(i, w, h)) for i, w, h in rec["images"]]
            record = {"images_path": list(range(len(images))), "question": rec["question"], "answers": rec["answers"],
                      "ques_type": rec["ques_type"], "options": rec["options"]}
            if got["raises"]:
                with pytest.raises(IndexError):
                    H.prepare_sample(record, setting["setting"], open_image=lambda i: images[i])
                continue
            s = H.prepare_sample(record, setting["setting"], open_image=lambda i: images[i])
            c = got["capture"]
            assert s.prompt == c["prompt"]
            assert [list(im.size) for im in s.vit_inputs] == c["vit_input_sizes"]
            assert s.n_image_tokens == c["n_image_tokens"]
            kw = H.generate_kwargs(128004)
            for k in ("pad_token_id", "eos_token_id", "max_new_tokens", "use_cache"):
                assert kw[k] == c["generate_kwargs"][k]
            assert c["generate_kwargs"]["pixel_values"] == [len(s.vit_inputs), 3, 364, 364]
            assert c["tokenizer_kwargs"] == {"return_tensors": "pt", "truncation": True, "max_length": H.MAX_PROMPT_TOKENS}
            row = H.result_row(record, s.question, got["result_row"]["raw"], len(s.vit_inputs))
            assert set(row) == set(got["result_row"])
            for k in ("gold", "raw", "question", "image_type", "multi_img"):
                assert row[k] == got["result_row"][k], k
        assert os.path.basename(H.shard_result_path("ckpt", 0, setting["setting"], "synth")) == setting["result_file"]


def test_split_shard_and_instruction():
    rows = list(range(17))
    parts = [H.split_shard(rows, i, 8) for i in range(8)]
    assert sum(parts, []) == rows and len(parts[0]) == 3
    assert H.get_instruction("direct", "multiple-choice").startswith("Answer with the option")
    assert H.get_instruction("none", "open-ended") == ""

"""Boundary check: the reference's evaluation script
/root/reference/evaluations/models/llava_multiimg_siglip_anyres.py is executed UNMODIFIED and in place
(run_llava_local_inference, EVAL:364-500) against leopard_amd.compat's model object.  Only the three loaders the script
calls are redirected (model -> our from_pretrained on a synthetic checkpoint; tokenizer / image processor -> local
stand-ins, since no tokenizer or processor files exist offline).  Kernels run on the CPU logic emulator at a micro
configuration; generated ids are checked against the CPU oracle.  Runs only where /root/reference exists (the build
container); the reference's files are never copied."""
import json
import os
import re
import sys
import types
from types import SimpleNamespace

import numpy as np
import pytest
import torch

REF = "/root/reference/evaluations/models"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def test_reference_harness_runs_unmodified(tmp_path):
    from PIL import Image
    from leopard_amd import compat
    from leopard_amd.checkpoint import save_synthetic_checkpoint
    from leopard_amd.synth import synth_image_u8, synth_state_dict_numpy
    from oracle import leopard_oracle as O
    from tests.emu_util import emu_ops
    from tests.test_emu_engine import micro_config

    for m in ("rouge", "editdistance"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["rouge"].Rouge = lambda *a, **k: SimpleNamespace(get_scores=lambda *a, **k: {"rouge-1": {"f": 0.0}, "rouge-l": {"f": 0.0}})
    sys.modules["editdistance"].eval = lambda a, b: 0 if a == b else max(len(a), len(b))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import llava_multiimg_siglip_anyres as L

    cfg = micro_config()
    ckpt = tmp_path / "ckpt"
    save_synthetic_checkpoint(str(ckpt), cfg)
    ops = emu_ops()
    IMG = "<|reserved_special_token_195|>"

    class Model(compat.LeopardForConditionalGeneration):
        def to(self, device):                      # the script asks for 'cuda:0'; this container has no GPU
            return super().to("cpu")

        def generate(self, *a, max_new_tokens=128, **k):      # the script asks for 128 new tokens; the emulator is slow
            return super().generate(*a, max_new_tokens=min(max_new_tokens, 6), **k)

    class Tok:
        pad_token_id = 0
        seen = []

        def __call__(self, texts, **kw):
            ids = []
            for piece in re.split("(" + re.escape(IMG) + ")", texts[0]):
                ids += [cfg.image_token_index] if piece == IMG else [1 + (ord(c) % 200) for c in piece[::7]]
            Tok.seen.append(ids)
            return {"input_ids": torch.tensor([ids])}

        def batch_decode(self, ids, **kw):
            Tok.seen.append(ids[0].tolist())
            return [" ".join(str(int(i)) for i in ids[0])]

    class Proc:
        def preprocess(self, image, return_tensors=None):
            return {"pixel_values": O.siglip_image_processor(image, size=cfg.vision_config.image_size)}

    L.myLlavaForConditionalGeneration.from_pretrained = classmethod(
        lambda cls, path, torch_dtype=None: Model.from_pretrained(path, torch_dtype=torch_dtype, ops=ops))
    L.AutoTokenizer.from_pretrained = staticmethod(lambda *a, **k: Tok())
    L.SiglipImageProcessor.from_pretrained = staticmethod(lambda *a, **k: Proc())

    (tmp_path / "models").mkdir()
    paths = []
    for i, (w, h) in enumerate([(400, 300), (800, 500)]):
        p = str(tmp_path / f"im{i}.png")
        Image.fromarray(synth_image_u8(i, w, h)).save(p)
        paths.append(p)
    recs = [{"images_path": paths[:1], "question": "<image> what?", "answers": ["x"], "ques_type": "open-ended", "options": None},
            {"images_path": paths, "question": "<image><image> which?", "answers": ["A"], "ques_type": "multiple-choice", "options": ["a", "b"]}]
    L.write_jsonl(str(tmp_path / "eval_synth.jsonl"), recs)
    cwd = os.getcwd()
    try:
        os.chdir(tmp_path / "models")
        L.run_llava_local_inference(SimpleNamespace(shard=0, num_shards=1, checkpoint=str(ckpt), dataset="synth",
                                                    setting="direct", view=False))
    finally:
        os.chdir(cwd)
    rows = [json.loads(l) for l in open(ckpt / "0_direct_synth_shard_details.jsonl")]
    assert len(rows) == 2 and set(rows[0]) == {"correct", "chosen", "gold", "raw", "question", "image_type", "multi_img", "correct_anls"}
    # the continuation the script decoded == the CPU oracle's greedy continuation on the same prompt + pixels
    from leopard_amd.harness import prepare_sample
    W = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    prompt_ids = [s for s in Tok.seen[0::2]]
    decoded = [s for s in Tok.seen[1::2]]
    for rec, ids, cont in zip(recs, prompt_ids, decoded):
        s = prepare_sample(rec, "direct")
        pix = torch.cat([O.siglip_image_processor(im, size=cfg.vision_config.image_size) for im in s.vit_inputs])
        assert ids.count(cfg.image_token_index) == len(s.vit_inputs)
        ref = O.greedy_generate(torch.tensor([ids]), pix, W, cfg, 6, eos_token_id=[128001, 128009])
        assert ref[0, len(ids):].tolist() == cont and len(cont) == 6

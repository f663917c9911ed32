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


def test_reference_idefics2_script_runs_unmodified(tmp_path):
    """evaluations/models/idefics2_multiimg.py (IDEF:33-131) executed in place against leopard_amd.idefics2_compat: the two names it
    imports from `transformers` are bound to our processor / model object, the tokenizer is a local
    stand-in, kernels run on the CPU emulator, `torch.device('cuda')` resolves to the CPU in this GPU-less container.  The
    generated ids the script decodes are checked against the Idefics2 CPU oracle."""
    import transformers
    from PIL import Image
    from safetensors.torch import save_file
    from leopard_amd import idefics2_compat as IC
    from leopard_amd.synth import idefics2_param_specs, idefics2_state_dict_numpy, synth_array, synth_image_u8
    from oracle import idefics2_oracle as IO
    from tests.emu_util import emu_ops
    from tests.test_emu_idefics2 import micro_idefics2
    from tests.test_idefics2_compat import ToyTokenizer

    for m in ("rouge", "editdistance"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["rouge"].Rouge = lambda *a, **k: SimpleNamespace(get_scores=lambda *a, **k: {"rouge-1": {"f": 0.0}, "rouge-l": {"f": 0.0}})
    sys.modules["editdistance"].eval = lambda a, b: 0 if a == b else max(len(a), len(b))
    if REF not in sys.path:
        sys.path.insert(0, REF)

    cfg = micro_idefics2()
    ops = emu_ops()
    ckpt = tmp_path / "ckpt"
    ckpt.mkdir()
    with open(ckpt / "config.json", "w") as f:
        json.dump(cfg.to_dict(), f)
    save_file({n: torch.from_numpy(synth_array(n, s, k)) for n, s, k in idefics2_param_specs(cfg)}, str(ckpt / "model.safetensors"))
    seen = {"prompts": [], "generated": []}

    class Model(IC.Idefics2ForConditionalGeneration):
        @classmethod
        def from_pretrained(cls, path, torch_dtype=torch.float16, **kw):
            m = super().from_pretrained(path, torch_dtype=torch_dtype, ops=ops)
            m.eos_token_id = (2,)
            return m

        def generate(self, *a, max_new_tokens=128, **k):          # the script asks for 128 new tokens; the emulator is slow
            seen["prompts"].append((k["input_ids"].clone(), k["pixel_values"].clone(), k["pixel_attention_mask"].clone()))
            out = super().generate(*a, max_new_tokens=min(max_new_tokens, 4), **k)
            seen["generated"].append(out[0].tolist())
            return out

    class Processor(IC.Idefics2Processor):
        @classmethod
        def from_pretrained(cls, path, **kw):
            p = super().from_pretrained(path, tokenizer=ToyTokenizer(), **kw)
            p.image_seq_len = cfg.perceiver_config.n_latents
            return p

    transformers.AutoProcessor, transformers.AutoModelForVision2Seq = Processor, Model        # the INTEGRATION.md binding
    sys.modules.pop("idefics2_multiimg", None)
    import idefics2_multiimg as I2

    class TorchOnCpu:                                   # the script says torch.device('cuda'); there is no GPU here
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*a, **k):
            return torch.device("cpu")
    I2.torch = TorchOnCpu()

    (tmp_path / "models").mkdir()
    paths = []
    for i, (w, h) in enumerate([(100, 60), (44, 58)]):
        p = str(tmp_path / f"im{i}.png")
        Image.fromarray(synth_image_u8(20 + i, w, h)).save(p)
        paths.append(p)
    recs = [{"images_path": paths[:1], "image_bytes": None, "question": "what?", "answers": ["x"], "options": None},
            {"images_path": paths, "image_bytes": None, "question": "<image> which?", "answers": ["A"], "options": ["a", "b"]}]
    I2.write_jsonl(str(tmp_path / "eval_synth.jsonl"), recs)
    cwd = os.getcwd()
    try:
        os.chdir(tmp_path / "models")
        I2.main(SimpleNamespace(shard=0, num_shards=1, checkpoint=str(ckpt), dataset="synth", setting="direct", resolution=56))
    finally:
        os.chdir(cwd)
    rows = [json.loads(l) for l in open(ckpt / "0_res56_direct_synth_shard_details.jsonl")]
    assert len(rows) == 2 and {"correct", "chosen", "gold", "raw", "question", "multi_img"} <= set(rows[0])
    assert rows[0]["multi_img"] is False and rows[1]["multi_img"] is True
    # the script prepends missing <image> tags: one image token run per image reached the model
    L_ = cfg.perceiver_config.n_latents
    assert [int((p[0] == 250).sum()) for p in seen["prompts"]] == [L_, 2 * L_]
    Wt = IO.weights_from_numpy(idefics2_state_dict_numpy(cfg))
    for (ids, pix, mask), got in zip(seen["prompts"], seen["generated"]):
        images = IC.Idefics2ForConditionalGeneration.unpad_images(pix, mask)
        want, cur = [int(t) for t in ids[0]], ids
        for _ in range(4):
            nxt = int(IO.prefill_logits(cur, images, Wt, cfg)[0, -1].argmax())
            want.append(nxt)
            if nxt == 2:
                break
            cur = torch.cat([cur, torch.tensor([[nxt]])], dim=1)
        assert got == want

"""The Idefics2 eval-script surface (leopard_amd/idefics2_compat.py): chat template, image-token expansion, processor tensors,
and generate() of the model object over the emulated kernels vs the CPU oracle."""
import numpy as np
import pytest
import torch
from PIL import Image

from leopard_amd import idefics2_compat as IC
from leopard_amd.idefics2 import Idefics2SynthSource
from leopard_amd.synth import idefics2_state_dict_numpy, synth_image_u8
from oracle import idefics2_oracle as IO
from tests.emu_util import emu_ops
from tests.test_emu_idefics2 import micro_idefics2


class ToyTokenizer:
    """Stands in for the checkpoint's tokenizer (third-party, no files offline): the three special strings are single ids,
    every other character is one id."""
    special = {IC.IMAGE_TOKEN: 250, IC.FAKE_TOKEN: 251, IC.END_OF_UTTERANCE: 252}

    def __call__(self, text, return_tensors="pt"):
        ids, i = [1], 0
        while i < len(text):
            for s, t in self.special.items():
                if text.startswith(s, i):
                    ids.append(t)
                    i += len(s)
                    break
            else:
                ids.append(3 + (ord(text[i]) % 200))
                i += 1
        t = torch.tensor([ids])
        return {"input_ids": t, "attention_mask": torch.ones_like(t)}

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(str(int(t)) for t in row if not (skip_special_tokens and int(t) in (1, 250, 251, 252))) for row in ids]


def test_chat_template_and_image_expansion():
    msgs = [{"role": "user", "content": [{"type": "text", "text": "<image><image>What is shown?"}]}]
    text = IC.apply_chat_template(msgs, add_generation_prompt=True)
    assert text == "User: <image><image>What is shown?<end_of_utterance>\nAssistant:"
    msgs2 = [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "hi"}]},
             {"role": "assistant", "content": [{"type": "text", "text": "yo"}]}]
    assert IC.apply_chat_template(msgs2) == "User:<image>hi<end_of_utterance>\nAssistant: yo<end_of_utterance>\n"
    ex = IC.expand_image_tokens("a<image><image>b<image>c", 2)
    F, I = IC.FAKE_TOKEN, IC.IMAGE_TOKEN
    assert ex == "a" + F + I * 2 + F + I * 2 + F + "b" + F + I * 2 + F + "c"


def test_processor_tensors():
    proc = IC.Idefics2Processor(ToyTokenizer(), longest_edge=56, image_seq_len=3)
    imgs = [Image.fromarray(synth_image_u8(1, 112, 60)), Image.fromarray(synth_image_u8(2, 30, 40))]
    out = proc(text="User: <image><image>q<end_of_utterance>\nAssistant:", images=imgs, return_tensors="pt")
    assert out["pixel_values"].shape == (1, 2, 3, 40, 56) and out["pixel_attention_mask"].shape == (1, 2, 40, 56)
    assert int(out["pixel_attention_mask"][0, 0].sum()) == 30 * 56 and int(out["pixel_attention_mask"][0, 1].sum()) == 40 * 30
    assert int((out["input_ids"] == 250).sum()) == 6 and int((out["input_ids"] == 251).sum()) == 3
    ref0 = IO.image_processor(imgs[0], 56)                       # oracle's restatement of the third-party image processor
    assert torch.equal(out["pixel_values"][0, 0, :, :30, :56], ref0)
    with pytest.raises(ValueError):
        proc(text="no image token", images=imgs)
    un = IC.Idefics2ForConditionalGeneration.unpad_images(out["pixel_values"], out["pixel_attention_mask"], "all")
    assert [tuple(u.shape) for u in un] == [(3, 30, 56), (3, 40, 30)]
    # the default is the reference's own rule (idefics_vlm_model.py:608, `> 0`): whole patches of the 40 x 56 canvas with any real pixel
    un = IC.Idefics2ForConditionalGeneration.unpad_images(out["pixel_values"], out["pixel_attention_mask"])
    assert [tuple(u.shape) for u in un] == [(3, 28, 56), (3, 28, 42)]


def test_generate_through_the_surface_matches_oracle():
    ops = emu_ops()
    cfg = micro_idefics2()
    model = IC.Idefics2ForConditionalGeneration(cfg, lambda dev, dt: Idefics2SynthSource(cfg, ops, dev, dt), torch.float16, ops,
                                                eos_token_id=(2,))
    model = model.to("cpu").eval()
    proc = IC.Idefics2Processor(ToyTokenizer(), longest_edge=cfg.longest_edge, image_seq_len=cfg.perceiver_config.n_latents)
    imgs = [Image.fromarray(synth_image_u8(5, 100, 60)), Image.fromarray(synth_image_u8(6, 44, 58))]
    msgs = [{"role": "user", "content": [{"type": "text", "text": "<image><image>ab"}]}]
    inputs = proc(text=proc.apply_chat_template(msgs, add_generation_prompt=True), images=imgs, return_tensors="pt")
    got = model.generate(**inputs, max_new_tokens=3)
    Wt = IO.weights_from_numpy(idefics2_state_dict_numpy(cfg))
    images = IC.Idefics2ForConditionalGeneration.unpad_images(inputs["pixel_values"], inputs["pixel_attention_mask"])
    ids = inputs["input_ids"]
    want = [int(t) for t in ids[0]]
    cur = ids
    for _ in range(3):
        logits = IO.prefill_logits(cur, images, Wt, cfg)
        nxt = int(logits[0, -1].argmax())
        want.append(nxt)
        if nxt == 2:
            break
        cur = torch.cat([cur, torch.tensor([[nxt]])], dim=1)
    assert got[0].tolist() == want


def test_patch_validity_any_through_the_surface_matches_oracle():
    """patch_validity='any' (transformers 4.4x rule): the smaller image keeps its partly padded patch column through the HIP path;
    first generated token == the oracle on the same crops, and the crops differ from the 'all' rule's."""
    ops = emu_ops()
    cfg = micro_idefics2()
    model = IC.Idefics2ForConditionalGeneration(cfg, lambda dev, dt: Idefics2SynthSource(cfg, ops, dev, dt), torch.float16, ops,
                                                eos_token_id=(2,)).to("cpu").eval()
    model.patch_validity = "any"
    proc = IC.Idefics2Processor(ToyTokenizer(), longest_edge=cfg.longest_edge, image_seq_len=cfg.perceiver_config.n_latents)
    imgs = [Image.fromarray(synth_image_u8(5, 100, 60)), Image.fromarray(synth_image_u8(6, 44, 58))]      # -> 33 x 56 and 56 x 42 px
    msgs = [{"role": "user", "content": [{"type": "text", "text": "<image><image>ab"}]}]
    inputs = proc(text=proc.apply_chat_template(msgs, add_generation_prompt=True), images=imgs, return_tensors="pt")
    crops_any = model.unpad_images(inputs["pixel_values"], inputs["pixel_attention_mask"], "any", 14)
    crops_all = model.unpad_images(inputs["pixel_values"], inputs["pixel_attention_mask"], "all", 14)
    assert [tuple(c.shape) for c in crops_any] != [tuple(c.shape) for c in crops_all]
    got = model.generate(**inputs, max_new_tokens=1)
    Wt = IO.weights_from_numpy(idefics2_state_dict_numpy(cfg))
    want = int(IO.prefill_logits(inputs["input_ids"], crops_any, Wt, cfg)[0, -1].argmax())
    assert int(got[0, -1]) == want

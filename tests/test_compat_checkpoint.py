"""The eval-script-facing model object + HF checkpoint ingest, on CPU over the emulated kernels: write a synthetic
checkpoint in the reference converter's key layout, load it through from_pretrained, and drive it with the exact call
sequence of the reference harness (EVAL:373-375, 448-454)."""
import numpy as np
import pytest
import torch

from leopard_amd import compat
from leopard_amd.checkpoint import CheckpointSource, load_config, save_synthetic_checkpoint
from leopard_amd.synth import synth_state_dict_numpy
from leopard_amd.tiler import siglip_normalize
from oracle import leopard_oracle as O
from tests.emu_util import emu_ops
from tests.test_emu_engine import micro_config


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    d = tmp_path_factory.mktemp("ckpt")
    cfg = micro_config()
    save_synthetic_checkpoint(str(d / "single"), cfg)
    save_synthetic_checkpoint(str(d / "sharded"), cfg, shard_bytes=2 << 20)
    return d, cfg


def test_checkpoint_roundtrip_and_sharding(ckpt):
    d, cfg = ckpt
    ref = synth_state_dict_numpy(cfg)
    assert load_config(str(d / "single")).to_dict() == cfg.to_dict()
    for sub in ("single", "sharded"):
        src = CheckpointSource(str(d / sub), "cpu", torch.float16)
        assert set(src.keys()) == set(ref)
        w = src.get("language_model.model.layers.1.mlp.down_proj.weight")
        assert w.dtype == torch.float16 and torch.equal(w.float(), torch.from_numpy(ref["language_model.model.layers.1.mlp.down_proj.weight"]))
        assert src.get("language_model.model.norm.weight").dtype == torch.float32
    import glob
    assert len(glob.glob(str(d / "sharded" / "model-*.safetensors"))) > 1


def test_reference_call_sequence(ckpt):
    d, cfg = ckpt
    ops = emu_ops()
    llava = compat.from_pretrained(str(d / "sharded"), torch_dtype=torch.float32, ops=ops)      # EVAL:373
    llava.eval()                                                                                # EVAL:374
    llava.to("cpu")                                                                             # EVAL:375 ('cuda:0' there)
    u8 = np.random.default_rng(4).integers(0, 256, (2, 28, 28, 3), dtype=np.uint8)
    images = torch.from_numpy(siglip_normalize(u8)).to(llava.device)                            # EVAL:403-405 (fp32 NCHW)
    input_ids = torch.tensor([[7, 250, 11, 250, 12]]).to(llava.device)
    attn_mask = (input_ids != 0).to(llava.device)
    out = llava.generate(input_ids, pixel_values=images, attention_mask=attn_mask, pad_token_id=0,
                         eos_token_id=[128001, 128009], max_new_tokens=3, use_cache=True)      # EVAL:448-452
    W = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    ref = O.greedy_generate(input_ids, images, W, cfg, 3, eos_token_id=[128001, 128009])
    assert out.shape == ref.shape and torch.equal(out, ref)
    res = llava(input_ids=input_ids, pixel_values=images, attention_mask=attn_mask, use_cache=True, return_dict=True)
    logits = O.prefill_logits(input_ids, images, W, cfg)
    assert res.logits.shape == logits.shape and (res.logits - logits).abs().max() <= 4e-3
    assert res.past_key_values.length == logits.shape[1]
    nxt = llava(input_ids=res.logits[:, -1:].argmax(-1), pixel_values=images, past_key_values=res.past_key_values)
    assert nxt.logits.shape == (1, 1, cfg.text_config.vocab_size)
    with pytest.raises(ValueError, match="number of image tokens"):
        llava.generate(input_ids, pixel_values=images[:1], max_new_tokens=1)


def test_harness_gpu_tiler_path_equals_host_path(ckpt, tmp_path):
    """harness.run_inference with the GPU tiler (emulated here) writes the same rows as with the reference's host pipeline:
    same prompt, same pixels (bit-exact tiles), same greedy continuation."""
    from PIL import Image
    from leopard_amd import harness
    from leopard_amd.gpu_tiler import GpuTiler
    from leopard_amd.synth import synth_image_u8
    d, cfg = ckpt
    ops = emu_ops()
    model = compat.LeopardForConditionalGeneration.from_pretrained(str(d / "single"), ops=ops).to("cpu").eval()

    class Tok:
        pad_token_id = 0

        def __call__(self, texts, **kw):
            ids = []
            for piece in texts[0].split(harness.TOK_IMG):
                ids += [1 + (ord(c) % 200) for c in piece[::9]] + [cfg.image_token_index]
            return {"input_ids": torch.tensor([ids[:-1]])}

        def batch_decode(self, ids, **kw):
            return [" ".join(str(int(i)) for i in ids[0])]

    paths = []
    for i, (w, h) in enumerate([(800, 500), (300, 300)]):          # thumbnail + 2 tiles, thumbnail only
        p = str(tmp_path / f"im{i}.png")
        Image.fromarray(synth_image_u8(40 + i, w, h)).save(p)
        paths.append(p)
    recs = [{"images_path": paths, "question": "<image><image> which?", "answers": ["A"], "ques_type": "open-ended", "options": None}]
    gen_kw = harness.generate_kwargs
    harness.generate_kwargs = lambda pad: {**gen_kw(pad), "max_new_tokens": 3}
    try:
        host = harness.run_inference(recs, model, Tok())
        dev = harness.run_inference(recs, model, Tok(), gpu_tiler=GpuTiler(ops, "cpu", out_size=cfg.vision_config.image_size))
    finally:
        harness.generate_kwargs = gen_kw
    assert host == dev and len(host) == 1 and host[0]["raw"] and host[0]["multi_img"]

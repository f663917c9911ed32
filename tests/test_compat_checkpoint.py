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


def _save_ckpt(path, cfg, transform):
    """A checkpoint of the micro model whose matrices are ``transform(name, synthetic fp32 values)`` (safetensors, the converter's key layout)."""
    import os
    from safetensors.torch import save_file
    from leopard_amd.synth import param_specs, synth_array
    os.makedirs(path, exist_ok=True)
    cfg.save(os.path.join(path, "config.json"))
    tensors = {}
    for n, s, k in param_specs(cfg):
        t = torch.from_numpy(synth_array(n, s, k))
        tensors[n] = transform(n, t) if t.dim() >= 2 else t
    save_file(tensors, os.path.join(path, "model.safetensors"))
    return tensors


def test_weights_that_are_not_fp16_exact(tmp_path):
    """the synthetic parameters are exact in fp16 by construction, a real checkpoint is not.  (a) bf16-VALUED weights (the
    released Leopard checkpoints are bf16-trained, train_multiimg_llava_siglip.sh:64) incl. values below fp16's normal range: exact down to 2^-17,
    below that bits are lost / flushed — counted by the loader, invisible in the logits; (b) fp32-VALUED weights: the cast is one rounding per
    weight that no precision mode corrects — its cost in the logits is printed beside the activation hand-overs' and the loader warns."""
    import warnings
    cfg = micro_config()
    ops = emu_ops()
    g = torch.Generator().manual_seed(5)

    def bf16_valued(name, t):
        w = (torch.randn(t.shape, generator=g) * 0.02).to(torch.bfloat16).float()
        flat = w.view(-1)
        flat[::97] = flat[::97] * 2.0 ** -12                                   # ~1 % of the weights far below 6.1e-5 (fp16's smallest normal)
        flat[::1013] = flat[::1013] * 2.0 ** -26                               # ... and a few that even fp16 subnormals cannot hold
        return flat.view(t.shape).to(torch.bfloat16).float()

    def fp32_valued(name, t):
        return torch.randn(t.shape, generator=g) * 0.02

    u8 = np.random.default_rng(4).integers(0, 256, (2, 28, 28, 3), dtype=np.uint8)
    images = torch.from_numpy(siglip_normalize(u8))
    input_ids = torch.tensor([[7, 250, 11, 250, 12, 31, 5]])
    out = {}
    for kind, tf in (("bf16", bf16_valued), ("fp32", fp32_valued)):
        tensors = _save_ckpt(str(tmp_path / kind), cfg, tf)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            m = compat.from_pretrained(str(tmp_path / kind), torch_dtype=torch.float32, ops=ops).to("cpu")
        st = m.weight_cast_stats
        got = m(input_ids=input_ids, pixel_values=images, return_dict=True).logits[0, -1]
        W_orig = {k: v.float() for k, v in tensors.items()}
        W_cast = {k: (v.to(torch.float16).float() if v.dim() >= 2 else v.float()) for k, v in tensors.items()}
        ref = O.prefill_logits(input_ids, images, W_orig, cfg, last_only=True)[0, 0]
        ref_cast = O.prefill_logits(input_ids, images, W_cast, cfg, last_only=True)[0, 0]
        scale = ref.abs().max().item()
        out[kind] = dict(stats=st, cast_cost=(ref_cast - ref).abs().max().item() / scale, total=(got - ref).abs().max().item() / scale,
                         vs_cast=(got - ref_cast).abs().max().item() / scale, warned=any("NOT exactly representable" in str(w.message) for w in rec))
        print(f"[weights {kind}-valued -> fp16] loader: {st['inexact_elements']} / {st['elements']} elements inexact, {st['flushed_to_zero']} flushed; "
              f"logits vs fp32-weight oracle: cast alone {out[kind]['cast_cost']:.2e}, engine {out[kind]['total']:.2e} (engine vs the cast-weight oracle {out[kind]['vs_cast']:.2e})")
    b, f = out["bf16"], out["fp32"]
    assert 0 < b["stats"]["inexact_elements"] < 0.02 * b["stats"]["elements"] and b["stats"]["flushed_to_zero"] > 0 and not b["warned"]
    assert b["cast_cost"] < 1e-5 and b["total"] < 1.2 * b["vs_cast"] + 1e-5              # the tiny values' lost bits do not show
    assert f["stats"]["inexact_elements"] > 0.9 * f["stats"]["elements"] and f["warned"]
    assert f["cast_cost"] > 20 * b["cast_cost"]                                            # a real rounding of every weight

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.  TEST INFRASTRUCTURE.

Runs only in the build container (needs /root/reference and the third-party ``transformers`` 5.15
installed there); its OUTPUT (small .json/.npz data files) is committed and travels to the GPU box,
the reference's Python never does.

What is executed from the reference (evaluations/models/llava_multiimg_siglip_anyres.py = EVAL):
  * EVAL:26-162   allocate_patches / select_best_resolution / resize_and_pad_image / divide_to_patches
  * EVAL:165-192  pixel_shuffle, myLlavaMultiModalProjector
  * EVAL:201-361  myLlavaForConditionalGeneration.forward, UNMODIFIED, called unbound over a shim object
                  whose sub-modules are third-party transformers SiglipVisionModel / LlamaForCausalLM
                   — the merge routine of transformers 4.38 is not available
                  offline and is restated in ``merge_438`` below.
  * EVAL:364-500  run_llava_local_inference, end-to-end with mocked model/tokenizer/processor loaders
                   to capture prompt strings, ViT-input order and generate kwargs.

Weights come from leopard_amd.synth (pure function of the parameter name), so fixtures only need to
store inputs + expected outputs.

usage:  python oracle/gen_golden.py            (rewrites tests/golden/*)
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import tempfile
import types
import warnings
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")
REF_MODELS = "/root/reference/evaluations/models"

warnings.filterwarnings("ignore")


def import_reference():
    class _Rouge:                               # scorers only; eval_utils instantiates Rouge at import
        def __init__(self, *a, **k):
            pass

        def get_scores(self, hyps, refs, avg=True):
            return {"rouge-1": {"f": 0.0}, "rouge-l": {"f": 0.0}}
    for m in ("rouge", "editdistance"):
        sys.modules[m] = types.ModuleType(m)
    sys.modules["rouge"].Rouge = _Rouge
    sys.modules["editdistance"].eval = lambda a, b: 0 if a == b else max(len(a), len(b))
    sys.path.insert(0, REF_MODELS)
    import llava_multiimg_siglip_anyres as L
    return L


# --------------------------------------------------------------------------------------------------
def gen_tiler(L):
    """(i) integer plans for seeded size lists + all BASELINE shapes + edge cases."""
    rng = np.random.default_rng(20240103)
    cases = [
        [(336, 336)], [(1344, 896)], [(896, 1344)], [(1344, 896)] * 4, [(1344, 896)] * 6, [(1344, 896)] * 8,
        [(1344, 896)] * 20, [(1344, 896)] * 49, [(364, 364)], [(363, 365)], [(546, 546)], [(545, 547)],
        [(910, 910)], [(182, 5000)], [(5000, 182)], [(8000, 6000)], [(100, 100), (4000, 3000)],
        [(728, 364)], [(364, 728)], [(2000, 2000)] * 3, [(1, 1)], [(1456, 1092)] * 2,
    ]
    for _ in range(200):
        n = int(rng.choice([1, 1, 2, 3, 4, 6, 8, 12, 20, 30, 48, 49]))
        cases.append([(int(rng.integers(50, 4200)), int(rng.integers(50, 4200))) for _ in range(n)])
    rows = []
    for sizes in cases:
        budget = 50 - len(sizes)
        alloc = L.allocate_patches(sizes, patch_budget=budget)
        res = [L.select_best_resolution(s, n) for s, n in zip(sizes, alloc)]
        tiles = [0 if r is None else (r[0] // 364) * (r[1] // 364) for r in res]
        rows.append({"sizes": sizes, "budget": budget, "allocate": alloc,
                     "resolution": [None if r is None else list(r) for r in res], "tiles": tiles})
    # select_best_resolution standalone sweep
    sweep = []
    for (w, h) in [(1344, 896), (896, 1344), (640, 480), (3000, 500), (500, 3000), (1000, 1000), (365, 365)]:
        for n in list(range(0, 21)) + [30, 49]:
            r = L.select_best_resolution((w, h), n)
            sweep.append({"size": [w, h], "n": n, "resolution": None if r is None else list(r)})
    # allocate with tight budgets (forces the scale branch)
    tight = []
    for _ in range(60):
        n = int(rng.integers(1, 12))
        sizes = [(int(rng.integers(300, 3000)), int(rng.integers(300, 3000))) for _ in range(n)]
        b = int(rng.integers(1, 30))
        tight.append({"sizes": sizes, "budget": b, "allocate": L.allocate_patches(sizes, patch_budget=b)})
    with open(os.path.join(OUT, "tiler_plans.json"), "w") as f:
        json.dump({"plans": rows, "resolution_sweep": sweep, "tight": tight}, f)
    print("tiler_plans.json:", len(rows), "plans")


def gen_tiles(L):
    """(ii) pixel work: letterbox + crop on seeded noise images (sha256 of every tile; one small case
    stored in full)."""
    from PIL import Image
    from leopard_amd.synth import synth_image_u8
    specs = [(0, 1344, 896, 8), (1, 896, 1344, 7), (2, 800, 500, 2), (3, 1000, 1000, 5), (4, 600, 1500, 4)]
    meta, arrays = [], {}
    for seed, w, h, n in specs:
        im = Image.fromarray(synth_image_u8(seed, w, h))
        res = L.select_best_resolution(im.size, n)
        padded = L.resize_and_pad_image(im, res)
        tiles = L.divide_to_patches(padded, 364) if padded is not None else []
        shas = [hashlib.sha256(np.asarray(t, dtype=np.uint8).tobytes()).hexdigest() for t in tiles]
        meta.append({"seed": seed, "w": w, "h": h, "n": n, "resolution": None if res is None else list(res),
                     "tile_sha256": shas})
        if seed == 2:
            arrays["tiles_seed2"] = np.stack([np.asarray(t, dtype=np.uint8) for t in tiles])
    with open(os.path.join(OUT, "tiles_meta.json"), "w") as f:
        json.dump(meta, f)
    np.savez_compressed(os.path.join(OUT, "tiles_seed2.npz"), **arrays)
    print("tiles:", [len(m["tile_sha256"]) for m in meta])


def gen_image_processor():
    """SiglipImageProcessor (third-party) on a thumbnail and a tile -> corner crops of pixel_values."""
    from PIL import Image
    from leopard_amd.synth import synth_image_u8
    try:
        from transformers import SiglipImageProcessor
        proc = SiglipImageProcessor(do_resize=True, size={"height": 364, "width": 364}, resample=3,
                                    do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
                                    image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
        ims = [Image.fromarray(synth_image_u8(7, 1344, 896)), Image.fromarray(synth_image_u8(8, 364, 364)),
               Image.fromarray(synth_image_u8(9, 336, 336))]
        outs = [np.asarray(proc.preprocess(im, return_tensors="pt")["pixel_values"])[0] for im in ims]
        np.savez_compressed(os.path.join(OUT, "image_processor.npz"),
                            crop0=outs[0][:, :48, :48], crop1=outs[1][:, 100:148, 200:248], crop2=outs[2][:, -48:, -48:],
                            mean=np.array([o.mean() for o in outs]), shape=np.array(outs[0].shape))
        print("image_processor.npz ok", type(proc).__name__)
    except Exception as e:                                      # pragma: no cover
        print("image processor fixture skipped:", repr(e))


def gen_pixel_shuffle_projector(L):
    """(iii)+(iv) reference pixel_shuffle on arange; reference projector at reduced dims."""
    from leopard_amd.config import tiny_config
    from leopard_amd.synth import synth_array, KIND_WEIGHT, KIND_BIAS
    x1 = torch.arange(2 * 16 * 8, dtype=torch.float32).reshape(2, 16, 8)
    x2 = torch.arange(1 * 676 * 4, dtype=torch.float32).reshape(1, 676, 4)
    ps1, ps2 = L.pixel_shuffle(x1), L.pixel_shuffle(x2)
    from transformers import LlavaConfig, SiglipVisionConfig, LlamaConfig
    vc = SiglipVisionConfig(hidden_size=64)
    tc = LlamaConfig(hidden_size=96, num_attention_heads=4, num_key_value_heads=2)
    cfg = LlavaConfig(vision_config=vc, text_config=tc, projector_hidden_act="gelu")
    proj = L.myLlavaMultiModalProjector(cfg).eval()
    sd = {"linear_1.weight": synth_array("multi_modal_projector.linear_1.weight", (96, 256), KIND_WEIGHT),
          "linear_1.bias": synth_array("multi_modal_projector.linear_1.bias", (96,), KIND_BIAS),
          "linear_2.weight": synth_array("multi_modal_projector.linear_2.weight", (96, 96), KIND_WEIGHT),
          "linear_2.bias": synth_array("multi_modal_projector.linear_2.bias", (96,), KIND_BIAS)}
    proj.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    feats = torch.from_numpy(np.random.default_rng(5).standard_normal((3, 16, 64)).astype(np.float32))
    with torch.no_grad():
        out = proj(feats)
    np.savez_compressed(os.path.join(OUT, "pixel_shuffle_projector.npz"),
                        ps_in1=x1.numpy(), ps_out1=ps1.numpy(), ps_out2=ps2.numpy(),
                        proj_in=feats.numpy(), proj_out=out.numpy())
    print("pixel_shuffle_projector.npz ok", tuple(out.shape))


# --------------------------------------------------------------------------------------------------
def merge_438(self, image_features, inputs_embeds, input_ids, attention_mask, labels):
    """Restatement of transformers 4.38..4.4x LlavaForConditionalGeneration.
    _merge_input_ids_with_image_features for right-padded / unpadded inputs with labels=None — the only
    case the eval script produces."""
    n_img, n_patch, dim = image_features.shape
    bsz, seq = input_ids.shape
    special = input_ids == self.config.image_token_index
    max_len = int(special.sum(-1).max()) * (n_patch - 1) + seq
    b_idx, t_idx = torch.where(~special)
    new_pos = torch.cumsum(special * (n_patch - 1) + 1, -1) - 1
    text_to = new_pos[b_idx, t_idx]
    emb = torch.zeros(bsz, max_len, dim, dtype=inputs_embeds.dtype)
    mask = torch.zeros(bsz, max_len, dtype=attention_mask.dtype)
    emb[b_idx, text_to] = inputs_embeds[b_idx, t_idx]
    mask[b_idx, text_to] = attention_mask[b_idx, t_idx]
    img_to = torch.full((bsz, max_len), True)
    img_to[b_idx, text_to] = False
    if img_to.sum() != image_features.shape[:-1].numel():
        raise ValueError("number of image tokens does not match number of images")
    emb[img_to] = image_features.reshape(-1, dim)
    mask |= img_to
    pos = (mask.cumsum(-1) - 1).masked_fill_(mask == 0, 1)
    self._captured_merge = (emb.clone(), mask.clone(), pos.clone())
    return emb, mask, None, pos


def build_shim(L, cfg):
    """Shim object exposing what EVAL:248-333 touches, with third-party sub-modules carrying the
    synthetic weights of leopard_amd.synth."""
    from transformers import LlavaConfig, SiglipVisionConfig, LlamaConfig, SiglipVisionModel, LlamaForCausalLM
    from leopard_amd.synth import synth_state_dict_numpy
    v, t = cfg.vision_config, cfg.text_config
    vc = SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                            num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                            image_size=v.image_size, patch_size=v.patch_size, layer_norm_eps=v.layer_norm_eps,
                            hidden_act=v.hidden_act)
    rp = {"rope_type": "llama3", "rope_theta": t.rope_theta, "factor": t.rope_scaling.factor,
          "low_freq_factor": t.rope_scaling.low_freq_factor, "high_freq_factor": t.rope_scaling.high_freq_factor,
          "original_max_position_embeddings": t.rope_scaling.original_max_position_embeddings}
    tc = LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                     num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                     num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size, pad_token_id=0,
                     rms_norm_eps=t.rms_norm_eps, max_position_embeddings=131072, rope_parameters=rp,
                     tie_word_embeddings=False, attn_implementation="eager")
    hf_cfg = LlavaConfig(vision_config=vc, text_config=tc, image_token_index=cfg.image_token_index,
                         projector_hidden_act=cfg.projector_hidden_act)
    sd = {k: torch.from_numpy(a) for k, a in synth_state_dict_numpy(cfg).items()}

    class Shim(nn.Module):
        def __init__(s):
            super().__init__()
            s.config = hf_cfg
            s.vision_tower = SiglipVisionModel(vc)
            s.language_model = LlamaForCausalLM(tc)
            s.multi_modal_projector = L.myLlavaMultiModalProjector(hf_cfg)

        def get_input_embeddings(s):
            return s.language_model.get_input_embeddings()
        _merge_input_ids_with_image_features = merge_438

    shim = Shim().eval()
    vkeys = set(shim.vision_tower.state_dict().keys())
    pre = "vision_tower.vision_model."
    vsd = {}
    for k, a in sd.items():
        if k.startswith(pre):
            kk = k[len(pre):]
            kk = kk if kk in vkeys else "vision_model." + kk
            vsd[kk] = a
    missing = shim.vision_tower.load_state_dict(vsd, strict=False)
    assert all("head" in m for m in missing.missing_keys), missing
    shim.language_model.load_state_dict({k[len("language_model."):]: a for k, a in sd.items()
                                         if k.startswith("language_model.")}, strict=True)
    shim.multi_modal_projector.load_state_dict({k[len("multi_modal_projector."):]: a for k, a in sd.items()
                                                if k.startswith("multi_modal_projector.")}, strict=True)
    return shim


def gen_tiny_e2e(L):
    """(v)+(vi) reference forward over the shim at the tiny config: logits, merged embeds, position ids,
    vision-tower output, projector output; plus greedy continuation computed with the same forward."""
    from leopard_amd.config import tiny_config
    cfg = tiny_config()
    shim = build_shim(L, cfg)
    rng = np.random.default_rng(11)
    cases = {}
    layouts = {
        "a": [1, 2, 500, 500, 3, 4, 5],
        "b": [7, 500, 8, 9, 500, 500, 500, 10, 11, 12, 13],
        "c": [500, 20, 21, 22, 500],
    }
    for name, ids in layouts.items():
        n_tiles = sum(1 for i in ids if i == 500)
        pix = rng.standard_normal((n_tiles, 3, 56, 56)).astype(np.float32)
        ids_t = torch.tensor([ids])
        with torch.no_grad():
            out = L.myLlavaForConditionalGeneration.forward(
                shim, input_ids=ids_t, pixel_values=torch.from_numpy(pix),
                attention_mask=torch.ones_like(ids_t, dtype=torch.bool), return_dict=True)
            vit = shim.vision_tower(torch.from_numpy(pix)).last_hidden_state
            vis = shim.multi_modal_projector(vit)
        emb, mask, pos = shim._captured_merge
        cases.update({f"{name}_ids": np.array(ids), f"{name}_pix": pix, f"{name}_logits": out.logits.numpy(),
                      f"{name}_vit": vit.numpy(), f"{name}_vis": vis.numpy(), f"{name}_embeds": emb.numpy(),
                      f"{name}_mask": mask.numpy().astype(np.int64), f"{name}_pos": pos.numpy()})
        # greedy continuation (4 tokens) by re-running the reference forward on the grown prompt
        cur = list(ids)
        for _ in range(4):
            t = torch.tensor([cur])
            with torch.no_grad():
                o = L.myLlavaForConditionalGeneration.forward(
                    shim, input_ids=t, pixel_values=torch.from_numpy(pix),
                    attention_mask=torch.ones_like(t, dtype=torch.bool), return_dict=True)
            cur.append(int(o.logits[0, -1].argmax()))
        cases[f"{name}_greedy"] = np.array(cur)
    # mismatch must raise
    try:
        t = torch.tensor([[1, 500, 2]])
        L.myLlavaForConditionalGeneration.forward(shim, input_ids=t, pixel_values=torch.zeros(2, 3, 56, 56),
                                                  attention_mask=torch.ones_like(t, dtype=torch.bool), return_dict=True)
        raised = False
    except ValueError:
        raised = True
    cases["mismatch_raises"] = np.array(int(raised))
    np.savez_compressed(os.path.join(OUT, "tiny_e2e.npz"), **cases)
    print("tiny_e2e.npz ok; logits", cases["a_logits"].shape, cases["b_logits"].shape, "raises", raised)


def gen_fullwidth_layers():
    """(vii) one full-width SigLIP layer and one full-width Llama layer (third-party modules, synthetic
    weights by name) on 32 / 48 tokens — outputs only; weights + inputs are regenerated from the seed."""
    from transformers import SiglipVisionConfig, LlamaConfig
    from transformers.models.siglip.modeling_siglip import SiglipEncoderLayer
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding
    from leopard_amd.config import full_config
    from leopard_amd.synth import param_specs, synth_array
    cfg = full_config()
    specs = {n: (s, k) for n, s, k in param_specs(cfg)}
    # --- SigLIP layer 0
    vc = SiglipVisionConfig(hidden_size=1152, intermediate_size=4304, num_hidden_layers=1, num_attention_heads=16,
                            image_size=364, patch_size=14, attn_implementation="eager")
    lay = SiglipEncoderLayer(vc).eval()
    pre = "vision_tower.vision_model.encoder.layers.0."
    lay.load_state_dict({k[len(pre):]: torch.from_numpy(synth_array(k, *specs[k])) for k in specs if k.startswith(pre)})
    x = torch.from_numpy(np.random.default_rng(21).standard_normal((2, 40, 1152)).astype(np.float32))
    with torch.no_grad():
        y = lay(x, attention_mask=None)
        y = y[0] if isinstance(y, tuple) else y
    # --- Llama layer 0
    rp = {"rope_type": "llama3", "rope_theta": 5e5, "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
          "original_max_position_embeddings": 8192}
    tc = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=1, num_attention_heads=32,
                     num_key_value_heads=8, vocab_size=512, rms_norm_eps=1e-5, max_position_embeddings=131072,
                     rope_parameters=rp, attn_implementation="eager")
    dl = LlamaDecoderLayer(tc, 0).eval()
    pre = "language_model.model.layers.0."
    dl.load_state_dict({k[len(pre):]: torch.from_numpy(synth_array(k, *specs[k])) for k in specs if k.startswith(pre)})
    rot = LlamaRotaryEmbedding(tc)
    S = 48
    xs = torch.from_numpy(np.random.default_rng(22).standard_normal((1, S, 4096)).astype(np.float32))
    pos = torch.arange(3000, 3000 + S).unsqueeze(0)          # large positions exercise the llama3 scaling
    cos, sin = rot(xs, pos)
    causal = torch.full((S, S), float("-inf")).triu(1)[None, None]
    with torch.no_grad():
        ys = dl(xs, attention_mask=causal, position_ids=pos, position_embeddings=(cos, sin))
        ys = ys[0] if isinstance(ys, tuple) else ys
    np.savez_compressed(os.path.join(OUT, "fullwidth_layers.npz"), siglip_out=y.numpy(), llama_out=ys.numpy(),
                        inv_freq=rot.inv_freq.numpy(), llama_pos0=np.array(3000))
    print("fullwidth_layers.npz ok", tuple(y.shape), tuple(ys.shape))


# --------------------------------------------------------------------------------------------------
def gen_harness_capture(L):
    """(viii) run_llava_local_inference end-to-end with mocked loaders; capture prompt / ViT-input order /
    generate kwargs / result row for synthetic records."""
    from PIL import Image
    from leopard_amd.synth import synth_image_u8
    tmp = tempfile.mkdtemp(prefix="leopard_cap_")
    os.makedirs(os.path.join(tmp, "models"))
    ckpt = os.path.join(tmp, "ckpt")
    os.makedirs(ckpt)

    def save_img(i, w, h):
        p = os.path.join(tmp, f"img_{i}_{w}x{h}.png")
        if not os.path.exists(p):
            Image.fromarray(synth_image_u8(i, w, h)).save(p)
        return p

    recs = []

    def rec(n_img, w, h, question, qt="open-ended", options=None, answers=("x",)):
        recs.append({"images_path": [save_img(i, w, h) for i in range(n_img)], "question": question,
                     "answers": list(answers), "ques_type": qt, "options": options})

    rec(1, 336, 336, "<image> What is shown?")
    rec(1, 1344, 896, "<image>\nDescribe the chart.")
    rec(2, 1344, 896, "<image><image> What is shown?")
    rec(6, 1344, 896, "<image>" * 6 + " Which slide mentions revenue?")
    rec(8, 1344, 896, "".join(f"Page {i}: <image>\n" for i in range(8)) + "Total?")
    rec(20, 1344, 896, "<image>" * 20 + " Summarise the deck.")
    rec(2, 800, 500, "<image> first <image> second. Which is larger?", "multiple-choice", ["left", "right"], ("A",))
    rec(3, 640, 480, "<image><image><image> Pick one.", "multiple-choice", ["a", "b", "c", "d"], ("C",))
    rec(1, 364, 364, "<image> tile-sized image")
    rec(1, 546, 546, "<image> rounds to 2x2?")
    rec(4, 1344, 896, "<image> <image> <image> <image> interleaved?")
    rec(2, 896, 1344, "Portrait pages <image><image>")
    rec(1, 2000, 300, "<image> wide strip")
    rec(1, 1344, 896, "<image> with \r\n\t\t\r\n\t\t odd whitespace")
    rec(3, 1344, 896, "<image><image><image> captioning case", "captioning", None, ("a caption",))
    # tag-count quirks: fewer tags than images -> too few image tokens (merge would raise);
    # more tags than images -> retain_n_images path
    rec(2, 1344, 896, "<image> only one tag for two images")
    rec(1, 1344, 896, "<image><image> two tags for one image")
    rec(2, 1344, 896, "no tag at all")
    rec(5, 1000, 1000, "<image>" * 5 + " squares")
    rec(12, 700, 900, "<image>" * 12 + " many")

    captures = []

    class FakeModel:
        device = torch.device("cpu")

        def eval(self):
            return self

        def to(self, *a, **k):
            return self

        def generate(self, input_ids, **kw):
            cap = captures[-1]
            cap["generate_kwargs"] = {k: (list(v.shape) if torch.is_tensor(v) else v) for k, v in kw.items()}
            cap["pixel_dtype"] = str(kw["pixel_values"].dtype)
            cap["n_input_ids"] = int(input_ids.shape[1])
            return torch.cat([input_ids, torch.tensor([[11, 12, 13]])], dim=1)

    class FakeTok:
        pad_token_id = 128004

        def __call__(self, texts, **kw):
            captures.append({"prompt": texts[0], "tokenizer_kwargs": {k: v for k, v in kw.items()},
                             "vit_input_sizes": list(FakeProc.sizes)})
            FakeProc.sizes = []
            n = texts[0].count("<|reserved_special_token_195|>")
            captures[-1]["n_image_tokens"] = n
            return {"input_ids": torch.arange(7 + n).unsqueeze(0)}

        def batch_decode(self, ids, **kw):
            return ["A fake answer"]

    class FakeProc:
        sizes = []

        def preprocess(self, image, return_tensors=None):
            FakeProc.sizes.append(list(image.size))
            return {"pixel_values": torch.zeros(1, 3, 364, 364)}

    L.myLlavaForConditionalGeneration.from_pretrained = classmethod(lambda cls, *a, **k: FakeModel())
    L.AutoTokenizer.from_pretrained = staticmethod(lambda *a, **k: FakeTok())
    L.SiglipImageProcessor.from_pretrained = staticmethod(lambda *a, **k: FakeProc())

    out = {"records": [], "settings": []}
    cwd = os.getcwd()
    try:
        os.chdir(os.path.join(tmp, "models"))
        for setting in ("direct", "cot", "none"):
            per_record = []
            for r in recs:                            # one record per run so that a record the reference
                captures.clear()                      # itself crashes on is captured as "raises"
                FakeProc.sizes = []
                L.write_jsonl(os.path.join(tmp, "eval_synth.jsonl"), [r])
                args = SimpleNamespace(shard=0, num_shards=1, checkpoint=ckpt, dataset="synth", setting=setting,
                                       view=False)
                try:
                    L.run_llava_local_inference(args)
                    rows = L.read_jsonl(os.path.join(ckpt, f"0_{setting}_synth_shard_details.jsonl"))
                    per_record.append({"capture": dict(captures[0]), "result_row": rows[0], "raises": None})
                except Exception as e:
                    per_record.append({"capture": None, "result_row": None, "raises": type(e).__name__})
            out["settings"].append({"setting": setting, "per_record": per_record,
                                    "result_file": f"0_{setting}_synth_shard_details.jsonl"})
    finally:
        os.chdir(cwd)
    for r in recs:                                    # strip tmp paths; keep the (index,w,h) recipe
        rr = dict(r)
        rr["images"] = [[int(os.path.basename(p).split("_")[1])] + [int(v) for v in
                        os.path.basename(p).split("_")[2][:-4].split("x")] for p in r["images_path"]]
        del rr["images_path"]
        out["records"].append(rr)
    with open(os.path.join(OUT, "harness_capture.json"), "w") as f:
        json.dump(out, f)
    print("harness_capture.json ok:", len(recs), "records x 3 settings")



def gen_idefics2_tiny():
    """Third-party Idefics2ForConditionalGeneration (the model class the reference's idefics2_multiimg.py loads, IDEF:27-29)
    at a tiny configuration with the seeded synthetic weights, on two images of different sizes padded to a common
    canvas with a pixel_attention_mask (what the reference's processor emits, IDEF:91-93)."""
    from transformers import Idefics2Config, Idefics2ForConditionalGeneration
    from leopard_amd.config import idefics2_tiny_config
    from leopard_amd.synth import idefics2_state_dict_numpy
    cfg = idefics2_tiny_config()
    v, t, pc = cfg.vision_config, cfg.text_config, cfg.perceiver_config
    hf_cfg = Idefics2Config(
        vision_config=dict(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                           num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size,
                           hidden_act=v.hidden_act, layer_norm_eps=v.layer_norm_eps),
        perceiver_config=dict(hidden_act="silu", hidden_size=t.hidden_size, rms_norm_eps=pc.rms_norm_eps,
                              resampler_n_latents=pc.n_latents, resampler_depth=pc.depth, resampler_n_heads=pc.n_heads,
                              resampler_head_dim=pc.head_dim, num_key_value_heads=pc.num_key_value_heads),
        text_config=dict(model_type="mistral", hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                         num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                         num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps,
                         rope_theta=t.rope_theta, sliding_window=t.sliding_window, max_position_embeddings=32768,
                         pad_token_id=0, head_dim=t.hidden_size // t.num_attention_heads),
        image_token_id=cfg.image_token_id, tie_word_embeddings=False, attn_implementation="eager")
    model = Idefics2ForConditionalGeneration(hf_cfg).eval()
    sd = {k: torch.from_numpy(a) for k, a in idefics2_state_dict_numpy(cfg).items()}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys, missing
    rng = np.random.default_rng(31)
    img_a = rng.standard_normal((3, 42, 56)).astype(np.float32)         # 3 x 4 patches
    img_b = rng.standard_normal((3, 58, 30)).astype(np.float32)         # 4 x 2 patches (+ remainder pixels)
    Hm, Wm = 58, 56
    pix = np.zeros((1, 2, 3, Hm, Wm), np.float32)
    msk = np.zeros((1, 2, Hm, Wm), np.int64)
    pix[0, 0, :, :42, :56] = img_a; msk[0, 0, :42, :56] = 1
    pix[0, 1, :, :58, :30] = img_b; msk[0, 1, :58, :30] = 1
    L = pc.n_latents
    ids = [5, 7] + [cfg.image_token_id] * L + [9, 11, 13] + [cfg.image_token_id] * L + [17, 19]
    ids_t = torch.tensor([ids])
    with torch.no_grad():
        out = model(input_ids=ids_t, attention_mask=torch.ones_like(ids_t), pixel_values=torch.from_numpy(pix),
                    pixel_attention_mask=torch.from_numpy(msk))
    np.savez_compressed(os.path.join(OUT, "idefics2_tiny.npz"), ids=np.array(ids), img_a=img_a, img_b=img_b,
                        logits=out.logits.numpy(), image_hidden_states=out.image_hidden_states.numpy())
    # The 4.4x patch-validity rule ("a patch belongs to the image when ANY of its pixels is real", `> 0`) on the same two images.
    # transformers 5.15 implements "ALL pixels real" (`== patch_size**2`); marking the partly padded patches of image b as fully
    # real in the mask makes the 5.15 model compute exactly what the `> 0` rule computes on the original mask: the same patch
    # grid (4 x 3 instead of 4 x 2), the same pixels (real columns 28-29 + zero padding), the same fractional position ids.
    msk_any = msk.copy()
    msk_any[0, 1, :56, :42] = 1
    with torch.no_grad():
        out_any = model(input_ids=ids_t, attention_mask=torch.ones_like(ids_t), pixel_values=torch.from_numpy(pix),
                        pixel_attention_mask=torch.from_numpy(msk_any))
    assert not np.allclose(out_any.image_hidden_states.numpy(), out.image_hidden_states.numpy())
    np.savez_compressed(os.path.join(OUT, "idefics2_tiny_any.npz"), ids=np.array(ids), pixel_values=pix, pixel_attention_mask=msk,
                        logits=out_any.logits.numpy(), image_hidden_states=out_any.image_hidden_states.numpy())
    # processor size rule sweep (third-party get_resize_output_image_size)
    from transformers.models.idefics2.image_processing_pil_idefics2 import get_resize_output_image_size
    from transformers.image_utils import SizeDict
    rows = []
    for (w, h) in [(1344, 896), (896, 1344), (980, 980), (400, 300), (2000, 100), (981, 5), (3000, 3000), (979, 1200)]:
        oh, ow = get_resize_output_image_size(np.zeros((3, h, w)), SizeDict(longest_edge=980, shortest_edge=0))
        rows.append([w, h, int(ow), int(oh)])
    with open(os.path.join(OUT, "idefics2_resize.json"), "w") as f:
        json.dump(rows, f)
    print("idefics2_tiny.npz ok; logits", out.logits.shape, "features", tuple(out.image_hidden_states.shape), rows[0])


def main():
    os.makedirs(OUT, exist_ok=True)
    L = import_reference()
    gen_tiler(L)
    gen_tiles(L)
    gen_image_processor()
    gen_pixel_shuffle_projector(L)
    gen_tiny_e2e(L)
    gen_fullwidth_layers()
    gen_harness_capture(L)
    gen_idefics2_tiny()


if __name__ == "__main__":
    main()

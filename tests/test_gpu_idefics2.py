"""-m gpu: Leopard-Idefics2 (NaViT SigLIP + perceiver + Mistral) through the C ABI vs the Idefics2 CPU oracle, full-width
layers at reduced depth, two images of different sizes (one is the BASELINE C4 image shape 1344x896 -> 980x653 ->
3220 patches); plus new-kernel-feature checks at production shapes (sliding window, head_dim 96 cross attention)."""
import numpy as np
import os

import pytest
import torch

from leopard_amd.config import idefics2_mid_config
from leopard_amd.synth import idefics2_state_dict_numpy, synth_image_u8

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from leopard_amd.ops import Ops
    return Ops()


def test_window_and_d96_kernels(ops):
    from tests.test_gpu_kernels import attn_ref, rnd
    dtype = torch.float16
    H, KV, D, S, Wn = 32, 8, 128, 700, 256
    qkv = rnd((S, (H + 2 * KV) * D), dtype, 65)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
    out = torch.empty(S, H * D, dtype=dtype, device=DEV)
    cu = torch.tensor([0, S], dtype=torch.int32, device=DEV)
    ops.attention(q, k, v, out, cu, cu, S, H, KV, D, D ** -0.5, True, True, window=Wn)
    qs = q.float().view(S, H, D).transpose(0, 1)
    ks = k.float().view(S, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
    vs = v.float().view(S, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
    i, j = torch.arange(S, device=DEV)[:, None], torch.arange(S, device=DEV)[None, :]
    sc = (qs @ ks.transpose(-1, -2) * D ** -0.5).masked_fill(~((j <= i) & (i - j < Wn)), float("-inf"))
    ref = (torch.softmax(sc, -1) @ vs).transpose(0, 1).reshape(S, H * D)
    assert (out.float() - ref).abs().max() <= 3e-3
    # perceiver cross attention: 64 latent queries x (3220 + 64) keys, 16 q / 4 kv heads x 96
    H, KV, D = 16, 4, 96
    cu_q, cu_k = [0, 64, 128], [0, 3284, 3284 + 1000 + 64]
    qq = rnd((128, H * D), dtype, 66)
    kv = rnd((cu_k[-1], 2 * KV * D), dtype, 67)
    for use_tr in (True, False):
        o = torch.full((128, H * D), float("nan"), dtype=dtype, device=DEV)
        ops.attention(qq, kv[:, :KV * D], kv[:, KV * D:], o, torch.tensor(cu_q, dtype=torch.int32, device=DEV),
                      torch.tensor(cu_k, dtype=torch.int32, device=DEV), 64, H, KV, D, D ** -0.5, False, use_tr)
        ref = attn_ref(qq, kv[:, :KV * D], kv[:, KV * D:], cu_q, cu_k, H, KV, D, D ** -0.5, False)
        assert (o.float() - ref).abs().max() <= 3e-3, use_tr


# tolerances = measured x 1.25 (profiles/r03_gpu_test_idefics2.txt: fp16 image features 1.14e-3 / logits 1.05e-3, bf16 7.12e-3 / 9.35e-3;
# normalised by the maximum magnitude), replacing the hand-picked 2.5e-3 / 2e-2 of round 2
@pytest.mark.parametrize("dtype,tol_f,tol_l", [(torch.float16, 1.43e-3, 1.32e-3), (torch.bfloat16, 8.9e-3, 1.17e-2)])
def test_idefics2_mid_prefill_vs_oracle(ops, dtype, tol_f, tol_l):
    from PIL import Image
    from leopard_amd.idefics2 import Idefics2Engine, Idefics2SynthSource, Idefics2Weights, preprocess_image_u8
    from oracle import idefics2_oracle as IO
    cfg = idefics2_mid_config()
    ims = [Image.fromarray(synth_image_u8(0, 1344, 896)), Image.fromarray(synth_image_u8(1, 500, 700))]
    u8 = [preprocess_image_u8(im, cfg.longest_edge) for im in ims]
    assert u8[0].shape == (653, 980, 3) and u8[1].shape == (700, 500, 3)
    L = cfg.perceiver_config.n_latents
    rng = np.random.default_rng(2)
    text = lambda n: rng.integers(3, 7000, n).tolist()
    ids = torch.tensor([text(6) + [cfg.image_token_id] * L + text(5) + [cfg.image_token_id] * L + text(20)])
    W = Idefics2Weights.build(cfg, Idefics2SynthSource(cfg, ops, torch.device(DEV), dtype), dtype)
    eng = Idefics2Engine(cfg, W, ops=ops, device=torch.device(DEV))
    res = eng.prefill(ids, [torch.from_numpy(a.copy()) for a in u8], all_logits=True, keep_parts=True)
    Wt = IO.weights_from_numpy(idefics2_state_dict_numpy(cfg))
    pix = [IO.image_processor(im, cfg.longest_edge) for im in ims]
    logits, parts = IO.prefill_logits(ids, pix, Wt, cfg, return_parts=True)
    f = (res.parts["image_features"].cpu() - parts["image_features"]).abs().max().item() / parts["image_features"].abs().max().item()
    a = (res.logits_all.cpu() - logits[0]).abs().max().item() / logits.abs().max().item()
    print(f"[idefics2 mid {dtype}] normalised-max error: image features {f:.2e}, logits {a:.2e} (|logit| max {logits.abs().max():.2f})")
    assert res.n_tiles == 2 and res.seq_len == ids.shape[1]
    assert f <= tol_f and a <= tol_l
    assert int(res.logits_last.argmax()) == int(logits[0, -1].argmax())


# Full depth (27 NaViT SigLIP + 3 perceiver + 32 Mistral layers) on BASELINE configs[3]'s sample (4 x 1344x896 -> 980x653, S = 312).
# Tolerances = measured x 1.25 (profiles/r03_error_growth_idefics2_c4.txt: logits 1.295e-3, image features 8.61e-4, normalised by the
# maximum magnitude), and the measured error must sit at the PREDICTED 16-bit hand-over budget (oracle.emulate_rounding: 1.454e-3; the
# per-layer table shows measured == predicted to < 1 % at every layer), i.e. the kernels add nothing beyond the operand roundings.
C4_FULL_TOL = {"logits": 1.62e-3, "features": 1.08e-3}


def test_idefics2_full_depth_c4_vs_oracle(ops):
    from leopard_amd.config import idefics2_full_config
    from leopard_amd.idefics2 import Idefics2Engine, Idefics2SynthSource, Idefics2Weights, preprocess_image_u8
    from oracle import idefics2_oracle as IO
    from oracle import leopard_oracle as O
    from tools.parity_report import idefics2_c4_sample
    cfg = idefics2_full_config()
    dtype, dev = torch.float16, torch.device(DEV)
    ims, ids = idefics2_c4_sample(cfg)
    src = Idefics2SynthSource(cfg, ops, dev, dtype)
    W = Idefics2Weights.build(cfg, src, dtype)
    eng = Idefics2Engine(cfg, W, ops=ops, device=dev)
    u8 = [torch.from_numpy(preprocess_image_u8(im, cfg.longest_edge).copy()) for im in ims]
    assert tuple(u8[0].shape) == (653, 980, 3) and ids.shape[1] == 312
    res = eng.prefill(ids, u8, keep_parts=True)
    got, feats = res.logits_last.float().cpu(), res.parts["image_features"].float().cpu()
    eng.precision = "lo4"                                  # the precision mode of Leopard-Idefics2 (tower + Mistral; connector fast)
    first4 = eng.prefill(ids, u8).logits_last.clone()
    got4 = first4.float().cpu()
    for rep in range(6):                                   # the text side (S = 312) runs the M-complete 384 x 128 ring of the correction phase: bit-reproducible
        assert torch.equal(eng.prefill(ids, u8).logits_last, first4), f"lo4 prefill repetition {rep} differs"
    del eng, W
    torch.cuda.empty_cache()
    # the fp32 oracle's logits and the rounding-emulating oracle's come from the COMMITTED fixture (tools/gen_idefics2_fixture.py, round 6: 2 x 23 TFLOP
    # of host arithmetic no longer recomputed on the GPU box at every run); the inputs are regenerated from their seeds and hash-checked against it
    import hashlib
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c4_idefics2_full_depth.npz"))
    assert np.array_equal(z["ids"], ids.numpy()), "C4 prompt synthesiser drifted from the fixture"
    assert hashlib.sha256(b"".join(np.ascontiguousarray(a.numpy()).tobytes() for a in u8)).digest() == z["images_sha256"].tobytes(), "C4 images drifted from the fixture"
    ref, emu = torch.from_numpy(z["logits_fp32"]), torch.from_numpy(z["logits_emu_fp16"])
    rows = z["feature_probe_rows"].tolist()
    feats = feats.reshape(-1, feats.shape[-1])[rows]
    parts = {"image_features": torch.from_numpy(z["feature_probe"])}
    feat_scale = float(z["feature_max_abs"][0])
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item() / scale
    pred = (emu - ref).abs().max().item() / scale
    ferr = (feats - parts["image_features"]).abs().max().item() / feat_scale
    print(f"[idefics2 C4 full depth fp16] logits {err:.3e} (predicted {pred:.3e}), image features (probe rows) {ferr:.3e}")
    assert err <= C4_FULL_TOL["logits"] and ferr <= C4_FULL_TOL["features"]
    assert 0.7 * pred <= err <= 1.4 * pred
    assert int(got.argmax()) == int(ref.argmax())
    err4 = (got4 - ref).abs().max().item() / scale
    print(f"[idefics2 C4 full depth fp16, lo4 correction] logits {err4:.3e} of the logit scale (max-abs {(got4 - ref).abs().max().item():.3e}): "
          f"north_star 1e-3 {'met' if err4 <= 1e-3 else 'x%.2f' % (err4 / 1e-3)}")
    assert err4 <= 1.0e-3 and int(got4.argmax()) == int(ref.argmax())

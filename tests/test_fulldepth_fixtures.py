"""-m "not gpu": integrity of the committed full-depth oracle fixtures (tools/gen_fulldepth_fixtures.py -> tests/golden/c{1,2,3}_full_depth.npz).

The fixtures hold 140 TFLOP of host arithmetic per run (C3), so they are not recomputed here; what IS checked on the CPU: the inputs
regenerate from their seeds to the fixture's SHA-256 / ids, the stored tensors are complete and finite, the emulated-rounding budget sits
where DESIGN.md 2.1 says, and the first traced tensor (SigLIP patch embedding + position table, which needs three parameter tensors only)
equals the oracle recomputed now on the probe rows — the fixture really is this oracle on these inputs."""
import os

import numpy as np
import pytest
import torch

from leopard_amd.config import full_config
from leopard_amd.synth import spec_table, synth_array
from tests.test_gpu_parity import FULL_CASES, FullDepthFixture, probe_rows

pytestmark = []


@pytest.mark.parametrize("case", ["c1", "c2", "c3"])
def test_fixture_integrity_and_first_trace_point(case):
    fx = FullDepthFixture(case)                                   # asserts meta, ids and the tile SHA-256
    cfg = full_config()
    V = cfg.text_config.vocab_size
    assert fx.ref.shape == (V,) and torch.isfinite(fx.ref).all()
    assert "fp16" in fx.emu and all(v.shape == (V,) and torch.isfinite(v).all() for v in fx.emu.values())
    assert fx.names[0] == "vit.embed" and fx.names[-1] == f"llm.{cfg.text_config.num_hidden_layers - 1}"
    assert len(fx.names) == 1 + cfg.vision_config.num_hidden_layers + 1 + cfg.text_config.num_hidden_layers
    # the 16-bit hand-over budget: fp16 logits 1e-3 .. 2e-3 of the logit scale at full depth, growing layer by layer
    d = (fx.emu["fp16"] - fx.ref).abs().max() / fx.ref.abs().max()
    assert 5e-4 < float(d) < 3e-3
    assert int(fx.emu["fp16"].argmax()) == int(fx.ref.argmax())
    tr = fx.pred["fp16"]
    assert tr["llm.31"] > tr["llm.0"] > 0 and tr["vit.26"] > tr["vit.0"] > 0
    # recompute the first trace point with the oracle
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    specs = spec_table(cfg)
    p = "vision_tower.vision_model.embeddings."
    W = {k: torch.from_numpy(synth_array(k, *specs[k])) for k in (p + "patch_embedding.weight", p + "patch_embedding.bias", p + "position_embedding.weight")}
    n_vit = FULL_CASES[case][3]
    tiles = sorted({0, n_vit - 1})
    x = O.siglip_embeddings(torch.from_numpy(siglip_normalize(fx.u8[tiles])), W, cfg)
    want = torch.stack([x[tiles.index(t), r] for t, r in probe_rows("vit.embed", (n_vit, 676, 1152))])
    assert torch.allclose(want, fx.probe["vit.embed"], rtol=0, atol=2e-6)


def test_idefics2_c4_fixture_integrity():
    """tests/golden/c4_idefics2_full_depth.npz (tools/gen_idefics2_fixture.py; round 6): the inputs regenerate from their seeds to the fixture's ids /
    SHA-256, the stored logits are complete and finite and the emulated 16-bit budget sits where DESIGN.md 2.1 says (1.3 - 1.5e-3 of the logit scale)."""
    import hashlib
    from leopard_amd.config import idefics2_full_config
    from leopard_amd.idefics2 import preprocess_image_u8
    from tools.parity_report import idefics2_c4_sample
    cfg = idefics2_full_config()
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c4_idefics2_full_depth.npz"))
    ims, ids = idefics2_c4_sample(cfg)
    u8 = [preprocess_image_u8(im, cfg.longest_edge) for im in ims]
    assert list(z["meta"]) == [4, 1344, 896, 312] and np.array_equal(z["ids"], ids.numpy())
    assert hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in u8)).digest() == z["images_sha256"].tobytes()
    ref, emu = torch.from_numpy(z["logits_fp32"]), torch.from_numpy(z["logits_emu_fp16"])
    assert ref.shape == emu.shape == (cfg.text_config.vocab_size,) and torch.isfinite(ref).all() and torch.isfinite(emu).all()
    d = float((emu - ref).abs().max() / ref.abs().max())
    assert 8e-4 < d < 2.5e-3 and int(emu.argmax()) == int(ref.argmax())
    assert z["feature_probe"].shape == (len(z["feature_probe_rows"]), cfg.text_config.hidden_size) and float(z["feature_max_abs"][0]) > 0

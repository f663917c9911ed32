"""Host logic of the round-5 precision surface (no kernels): which schedule a ``from_pretrained`` request resolves to, where the run-info side
file goes, and that the continuous-batching harness prepares a record's pixels only when a slot admits it."""
import json
import os

import pytest
import torch

from leopard_amd import compat, harness, reference_shim


def test_resolve_precision(monkeypatch):
    monkeypatch.delenv("LEOPARD_AMD_PRECISION", raising=False)
    f16, bf16, f32 = torch.float16, torch.bfloat16, torch.float32
    assert compat.resolve_precision(f32, f16) == "lo4"               # EVAL:373 asks for fp32: the mode that meets 1e-3 against fp32 arithmetic
    assert compat.resolve_precision(None, f16) == "lo4"
    assert compat.resolve_precision(f16, f16) == "fast"              # a 16-bit request is served as asked
    assert compat.resolve_precision(f32, bf16) == "fast"             # no bf16 schedule meets the figure: no pretence
    assert compat.resolve_precision(f32, f16, "split") == "split"
    assert compat.resolve_precision(f32, f16, "split", tp_size=8) == "fast"
    assert compat.resolve_precision(f32, f16, tp_size=8) == "lo4"    # tensor-parallel engines run lo4 too (round 5)
    monkeypatch.setenv("LEOPARD_AMD_PRECISION", "fast")
    assert compat.resolve_precision(f32, f16, "lo4") == "fast"       # the environment overrides everything
    monkeypatch.setenv("LEOPARD_AMD_PRECISION", "exact")
    with pytest.raises(ValueError):
        compat.resolve_precision(f32, f16)


def test_run_info_goes_next_to_the_result_shards(tmp_path, monkeypatch):
    monkeypatch.delenv("LEOPARD_AMD_RUN_INFO_DIR", raising=False)
    ck, cwd = tmp_path / "ckpt", tmp_path / "cwd"
    ck.mkdir(); cwd.mkdir()
    monkeypatch.chdir(cwd)
    with pytest.warns(UserWarning, match="precision mode lo4"):
        reference_shim._record_run_info("LlavaForConditionalGeneration", str(ck), torch.float32, torch.float16, "lo4")
    info = json.load(open(ck / "leopard_amd_run_info.json"))          # EVAL:496-497: the shard files live in the checkpoint directory
    assert info["requested_torch_dtype"] == "float32" and info["compute_dtype"] == "float16" and info["precision_mode"].startswith("lo4")
    assert not os.path.exists(cwd / "leopard_amd_run_info.json")
    other = tmp_path / "elsewhere"
    other.mkdir()
    monkeypatch.setenv("LEOPARD_AMD_RUN_INFO_DIR", str(other))
    reference_shim._record_run_info("X", str(ck), torch.float16, torch.float16, "fast")
    assert json.load(open(other / "leopard_amd_run_info.json"))["precision_mode"].startswith("fast")
    monkeypatch.delenv("LEOPARD_AMD_RUN_INFO_DIR")
    reference_shim._record_run_info("X", str(tmp_path / "not_a_dir"), torch.float16, torch.float16, "fast")     # no such directory: the working directory
    assert os.path.exists(cwd / "leopard_amd_run_info.json")


def test_streaming_harness_prepares_pixels_when_a_slot_admits_the_record(tmp_path):
    """run_inference(batch_size > 1) used to tile and upload EVERY record before decoding.  Now only plan + prompt + ids are made
    up front; the pixel closure of a record runs when generate_stream asks for it."""
    from PIL import Image
    import numpy as np
    paths = []
    for i in range(5):
        p = str(tmp_path / f"im{i}.png")
        Image.fromarray(np.full((300 + 10 * i, 400, 3), 30 * i, dtype=np.uint8)).save(p)
        paths.append(p)
    recs = [{"images_path": [paths[i]], "question": "<image> what?", "answers": ["x"], "ques_type": "open-ended", "options": None} for i in range(5)]

    class Tok:
        pad_token_id = 0

        def __call__(self, texts, return_tensors=None, truncation=None, max_length=None):
            return {"input_ids": torch.tensor([[1 + (ord(c) % 50) for c in texts[0][:40]]])}

        def batch_decode(self, ids, skip_special_tokens=True):
            return ["ok"]
    events = []

    class Model:
        device = torch.device("cpu")

        def generate_stream(self, requests, batch_size=8, eos_token_id=None, max_new_tokens=128, stats=None, **kw):
            events.append(("stream_called_with_callables", all(callable(px) for _, px in requests)))
            outs = []
            for ids, px in requests:                                   # "admit" one at a time
                t = px()
                events.append(("pixels", tuple(t.shape)))
                outs.append(torch.cat([ids, torch.tensor([[7]])], dim=1))
            return outs
    rows = harness.run_inference(recs, Model(), Tok(), batch_size=2)
    assert len(rows) == 5 and all(r["raw"] == "ok" for r in rows)
    assert events[0] == ("stream_called_with_callables", True)
    assert sum(1 for e in events if e[0] == "pixels") == 5
    # the plan made without decoding any pixel agrees with the full preparation
    q, n_vit, enc = harness.plan_record(recs[3], "direct", Tok())
    s = harness.prepare_sample(recs[3], "direct")
    assert q == s.question and n_vit == len(s.vit_inputs)


def test_ops_refuse_a_16_bit_bias():
    """The C-ABI reads biases / addmats as const float*: the wrappers refuse anything else instead of letting the kernel read past the tensor."""
    from tests.emu_util import emu_ops
    ops = emu_ops()
    a = torch.zeros(64, 128, dtype=torch.float16)
    w = torch.zeros(128, 128, dtype=torch.float16)
    out = torch.zeros(64, 128, dtype=torch.float16)
    with pytest.raises(TypeError, match="float32"):
        ops.gemm(a, w, out, bias=torch.zeros(128, dtype=torch.float16))
    ops.gemm(a, w, out, bias=torch.zeros(128, dtype=torch.float32))

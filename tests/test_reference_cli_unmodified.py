"""Row (b) of SURVEY.md section 8: the reference's evaluation script run from ITS OWN command line, unmodified and in place,
through the two shipped entry points — no test-local patching of the script or of the model classes:

    python -m leopard_amd.run_reference_eval /root/reference/evaluations/models/llava_multiimg_siglip_anyres.py -- <its CLI>
    PYTHONPATH=<repo>/leopard_amd/hf_shim:<repo> python /root/reference/evaluations/models/llava_multiimg_siglip_anyres.py <its CLI>

(the second is what run_eval_llava_siglip_multiimg.sh:9-11 does per GPU, with one environment variable added).  The tokenizer and
the image processor are the REAL third-party ones, loaded from files this test writes (a character-level tokenizer.json with the
reference's special tokens, a SigLIP preprocessor_config.json); the checkpoint is a synthetic one in the converter's layout.
Kernels run on the CPU logic emulator at a micro configuration (smoke-run knobs LEOPARD_AMD_LIB / _FORCE_DEVICE /
_MAX_NEW_TOKENS).  The result rows are compared with leopard_amd.harness driven over the same records, and the generated ids with
the CPU oracle.  Runs only where /root/reference exists (the build container)."""
import json
import os
import subprocess
import sys

import pytest
import torch

REF = "/root/reference/evaluations/models"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
IMG, RST20, RST21 = "<|reserved_special_token_195|>", "<|reserved_special_token_20|>", "<|reserved_special_token_21|>"


def write_tokenizer(path, image_token_index):
    """A real `tokenizers` tokenizer: one token per character (ids 1..), unk = 0, then the three special tokens the reference's
    prompt uses, the image token landing exactly on cfg.image_token_index."""
    from tokenizers import Regex, Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    chars = [chr(c) for c in range(32, 127)] + ["\n", "\t", "\r"]
    vocab = {"<unk>": 0}
    for c in chars:
        vocab[c] = len(vocab)
    while len(vocab) < image_token_index:
        vocab[f"<filler_{len(vocab)}>"] = len(vocab)
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split(Regex("."), behavior="isolated")
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", pad_token="<unk>")
    fast.add_special_tokens({"additional_special_tokens": [IMG, RST20, RST21]})
    assert fast.convert_tokens_to_ids(IMG) == image_token_index
    fast.save_pretrained(path)


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    from PIL import Image
    from leopard_amd.checkpoint import save_synthetic_checkpoint
    from leopard_amd.synth import synth_image_u8
    from tests.emu_util import EMU, emu_ops
    from tests.test_emu_engine import micro_config
    emu_ops()                                                   # make sure the emulator library is built
    root = tmp_path_factory.mktemp("refcli")
    cfg = micro_config()
    ckpt = root / "ckpt"
    save_synthetic_checkpoint(str(ckpt), cfg, shard_bytes=1 << 20)          # sharded safetensors + index, converter key layout
    write_tokenizer(str(ckpt), cfg.image_token_index)
    models = root / "models"
    models.mkdir()
    proc = models / "siglip-so400m-14-364-flash-attn2-navit"              # the relative path EVAL:378 loads the processor from
    proc.mkdir()
    S = cfg.vision_config.image_size
    json.dump({"image_processor_type": "SiglipImageProcessor", "do_resize": True, "size": {"height": S, "width": S}, "resample": 3,
               "do_rescale": True, "rescale_factor": 1 / 255, "do_normalize": True, "image_mean": [0.5, 0.5, 0.5],
               "image_std": [0.5, 0.5, 0.5], "do_convert_rgb": None}, open(proc / "preprocessor_config.json", "w"))
    paths = []
    for i, (w, h) in enumerate([(400, 300), (800, 500)]):
        p = str(root / f"im{i}.png")
        Image.fromarray(synth_image_u8(i, w, h)).save(p)
        paths.append(p)
    recs = [{"images_path": paths[:1], "question": "<image> what?", "answers": ["x"], "ques_type": "open-ended", "options": None},
            {"images_path": paths, "question": "<image><image> which?", "answers": ["A"], "ques_type": "multiple-choice", "options": ["a", "b"]}]
    with open(root / "eval_synth.jsonl", "w") as f:
        for r in recs:
            f.write(json.dumps(r) + "\n")
    env = dict(os.environ, LEOPARD_AMD_LIB=EMU, LEOPARD_AMD_FORCE_DEVICE="cpu", LEOPARD_AMD_MAX_NEW_TOKENS="4", CUDA_VISIBLE_DEVICES="")
    return root, cfg, ckpt, models, recs, env


def check_rows(ckpt, cfg, recs):
    rows = [json.loads(l) for l in open(ckpt / "0_direct_synth_shard_details.jsonl")]
    assert len(rows) == 2
    assert set(rows[0]) == {"correct", "chosen", "gold", "raw", "question", "image_type", "multi_img", "correct_anls"}
    # our own harness over the same records writes the same rows (prompt construction, tiler, generate, scoring inputs)
    from leopard_amd import harness
    from leopard_amd.reference_shim import _llava_class
    from tests.emu_util import emu_ops
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(str(ckpt))
    model = _llava_class().from_pretrained(str(ckpt), ops=emu_ops()).to("cpu")
    os.environ["LEOPARD_AMD_MAX_NEW_TOKENS"] = "4"
    try:
        mine = harness.run_inference(recs, model, tok, "direct")
    finally:
        del os.environ["LEOPARD_AMD_MAX_NEW_TOKENS"]
    for a, b in zip(mine, rows):
        assert a["raw"] == b["raw"] and a["question"] == b["question"] and a["multi_img"] == b["multi_img"] and a["gold"] == b["gold"]
    os.remove(ckpt / "0_direct_synth_shard_details.jsonl")


def test_launcher_module_runs_the_reference_cli(workdir):
    root, cfg, ckpt, models, recs, env = workdir
    env = dict(env, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-m", "leopard_amd.run_reference_eval", os.path.join(REF, "llava_multiimg_siglip_anyres.py"), "--",
                        "--shard", "0", "--num_shards", "1", "-c", str(ckpt), "-d", "synth", "-s", "direct"],
                       cwd=str(models), env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "saving to" in r.stdout
    check_rows(ckpt, cfg, recs)


def test_pythonpath_shim_runs_the_reference_cli(workdir):
    """The reference's own command line (run_eval_llava_siglip_multiimg.sh:10) with ONE environment variable added."""
    root, cfg, ckpt, models, recs, env = workdir
    env = dict(env, PYTHONPATH=os.path.join(REPO, "leopard_amd", "hf_shim") + os.pathsep + REPO)
    r = subprocess.run([sys.executable, os.path.join(REF, "llava_multiimg_siglip_anyres.py"),
                        "--shard", "0", "--num_shards", "1", "-c", str(ckpt), "-d", "synth", "-s", "direct"],
                       cwd=str(models), env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    check_rows(ckpt, cfg, recs)

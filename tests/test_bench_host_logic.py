"""Host-side logic of bench.py that needs no GPU: the per-launch roofline timer's handling of repeated passes, the algorithmic FLOP count
of the C3 step, the workload labels."""
import types

import pytest
import torch

import bench


class _Ev:
    def __init__(self, ms):
        self.ms = ms

    def elapsed_time(self, other):
        return other.ms


def _timer(times_per_pass, passes, shapes=None):
    t = bench.GemmTimer()
    t.passes = passes
    n = len(times_per_pass[0])
    shapes = shapes or [(100 + i, 128, 64) for i in range(n)]
    for p in range(passes):
        for i in range(n):
            t.records.append((2.0 * shapes[i][0] * shapes[i][1] * shapes[i][2], _Ev(0.0), _Ev(times_per_pass[p][i]), shapes[i]))
    return t


def test_gemm_timer_takes_each_launch_at_its_fastest_pass(monkeypatch):
    """An event pair also spans any moment the host fell behind the device: with two passes over the same launch sequence a launch's time is
    the minimum over the passes, so one stall is not booked as GEMM time; unequal sequences are summed as they are."""
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    t = _timer([[1.0, 2.0, 40.0], [1.1, 1.9, 3.0]], 2)             # the third launch of pass 0 carries a 37 ms stall
    flops, ms, n = t.summary()
    assert n == 6 and ms == pytest.approx(2 * (1.0 + 1.9 + 3.0))
    dom = t.dominant()
    assert dom["shape_MNK"] == [102, 128, 64] and dom["launches"] == 2 and dom["avg_launch_ms"] == pytest.approx(3.0)
    one = _timer([[1.0, 2.0, 40.0]], 1)
    assert one.summary()[1] == pytest.approx(43.0)
    odd = bench.GemmTimer()
    odd.passes = 2
    for ms_, shape in ((1.0, (1, 128, 64)), (2.0, (2, 128, 64)), (5.0, (3, 128, 64)), (7.0, (4, 128, 64))):     # not the same sequence twice
        odd.records.append((1.0, _Ev(0.0), _Ev(ms_), shape))
    assert odd.summary()[1] == pytest.approx(15.0)


def test_algorithmic_flops_of_the_c3_step():
    """140.14 TFLOP per C3 sample (42 ViT inputs, S = 7187) — the figure every 'of peak' number in the bench line is priced with."""
    from leopard_amd.config import full_config
    f = bench.algorithmic_flops(full_config(), 42, 7187)
    assert f["total"] / 1e12 == pytest.approx(140.14, abs=0.01)
    assert f["total"] == 42 * (f["vit"] // 42 + f["projector"] // 42) + f["llm_linear"] + f["llm_attention"] + f["lm_head_last"]


def test_fp8_detail_names_the_attention_arithmetic():
    on, off = types.SimpleNamespace(fp8_attention=1), types.SimpleNamespace(fp8_attention=0)
    assert "f8f6f4" in bench.fp8_detail(on) and "f16 SigLIP attention" in bench.fp8_detail(on)
    assert bench.fp8_detail(off) == bench.FP8_DETAIL and "f16 attention" in bench.FP8_DETAIL


class _FakeD:
    @staticmethod
    def max_over_ranks(v, dev):
        return v


def _tp_args(**kw):
    d = dict(steps=2, warmup=1, dtype="f16", images=6, width=1344, height=896, precision="lo4", no_fuse=False, opt=[], tp_timeout=5.0)
    d.update(kw)
    return types.SimpleNamespace(**d)


@pytest.mark.parametrize("case", ["ok", "crash", "hang", "silent"])
def test_tp_child_failures_cost_the_tp_object_not_the_line(monkeypatch, tmp_path, case):
    """bench.run_tp_child: the N > 1 tensor-parallel measurement is a child process per rank with its own rendezvous port; whatever the child does
    (result, non-zero exit, hang, no output) the parent gets a dict back — the headline line is printed either way."""
    import subprocess
    import sys
    body = {"ok": "import json, os; assert os.environ['MASTER_PORT'].isdigit() and 'TORCHELASTIC_RUN_ID' not in os.environ; print(json.dumps({'tp': {'value': 1.5, 'n_gpus': 2}}))",
            "crash": "import sys; sys.stderr.write('RCCL abort'); sys.exit(134)",
            "hang": "import time; time.sleep(60)",
            "silent": "pass"}[case]
    script = tmp_path / "child.py"
    script.write_text(body)
    real_popen = subprocess.Popen
    seen = {}

    def fake_popen(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        return real_popen([sys.executable, str(script)], **kw)
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "x")
    res = bench.run_tp_child(_tp_args(tp_timeout=3.0 if case == "hang" else 30.0), 0, 2, _FakeD, None)
    assert "--parallelism" in seen["cmd"] and seen["cmd"][seen["cmd"].index("--parallelism") + 1] == "tp" and "--gpus" in seen["cmd"]
    assert seen["env"]["MASTER_PORT"].isdigit() and not any(k.startswith("TORCHELASTIC_") for k in seen["env"])
    if case == "ok":
        assert res == {"value": 1.5, "n_gpus": 2}
    else:
        assert "error" in res
        if case == "crash":
            assert "134" in res["error"] and "RCCL abort" in res["stderr_tail"]
        if case == "hang":
            assert "did not finish" in res["error"]


def test_library_load_imports_torch_first():
    """One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64, the library is linked against the system one, and whichever is
    loaded first serves both.  _lib.load() therefore imports torch BEFORE it dlopens the library (round 6: build() followed by smoke() in one
    process loaded them the other way round and the first launch on the GPU box said "no ROCm-capable device").  Checked in a fresh
    interpreter, on the order in which the two shared objects appear in the process's memory map."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys\n"
            "from leopard_amd import _lib\n"
            "assert 'torch' not in sys.modules\n"
            "_lib.load()\n"
            "assert 'torch' in sys.modules\n"
            "maps = open('/proc/self/maps').read()\n"
            "hip = [l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l]\n"
            "assert hip and len(set(hip)) == 1, set(hip)\n"                 # a single HIP runtime in the process ...
            "assert '/torch/' in hip[0], hip[0]\n"                          # ... and it is the one torch was built with
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=repo, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr

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

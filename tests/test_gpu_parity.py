"""-m gpu: the HIP prefill path vs the CPU oracle on identical seeded inputs.

  * mid configuration (full width, 2 + 2 layers, S = 566): every intermediate (ViT features, visual tokens, merged embeddings,
    all-position and last-position logits), greedy continuation, error conventions;
  * FULL depth and width (27 SigLIP + 32 Llama-3.1-8B layers) on C1 (1 x 336x336, S = 228) and on C2 (1 x 1344x896 -> 7 ViT
    inputs, S = 1242), fp16 and bf16, last-position logits vs the fp32 oracle and vs the oracle with the kernels' 16-bit
    hand-over roundings emulated (oracle.emulate_rounding);
  * the C3 SEQUENCE LENGTH (6 x 1344x896 -> 42 ViT inputs, S = 7187) at reduced depth (2 + 2 full-width layers): the residual
    stream of every one of the 7187 rows and the logits vs the oracle — the 29-row-tile GEMMs and the 7187-key softmax meet
    the oracle directly, not only through self-consistency properties (those stay, at the end of the file).

Tolerance (north_star: "logits within 1e-3 fp16").  The oracle is the reference's fp32 CPU arithmetic; the HIP path feeds
16-bit operands to the MFMAs (fp32 accumulate, fp32 residual stream, fp32 last-token head), so it differs by one rounding of
2^-11 (fp16) / 2^-8 (bf16) relative at every kernel hand-over.  The error is asserted on the logits NORMALISED by the logit
scale, max|diff| / max|logit| (an absolute 1e-3 on logits of magnitude ~6 is finer than the fp16 grid of the values themselves).
The FAST schedule's bounds below are the MEASURED errors x 1.25 (profiles/r02_error_growth.txt holds the per-layer table and the predicted
budget from the rounding-emulating oracle, which the measurements match); fp16 meets the 1e-3 at the depths the reference's
C1 / C2 cases have where stated, and where it does not the table shows the same excess for the emulated oracle: it is the
price of 16-bit operands at that depth, not of the kernels.  Integer / index work is bit-exact."""
import os

import numpy as np
import pytest
import torch

from leopard_amd.config import full_config, mid_config
from leopard_amd.synth import synth_image_u8, synth_prompt_ids, synth_state_dict_numpy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# FAST schedule: max|diff| / max|logit| bounds = measured x 1.25 on the fused norm / RoPE schedule (measured values in the comments, the
# per-layer table and the predicted budget in profiles/r02_error_growth.txt).  The lo4 and split schedules are asserted at north_star's 1e-3
# (test_full_depth_lo4_meets_1e_3, test_full_depth_split_operands_meets_1e_3), not at a multiple of what was measured.
LOGIT_TOL = {torch.float16: 1.25e-3, torch.bfloat16: 1.03e-2}     # mid configuration, 2 + 2 layers: all-position logits 9.9e-4 / 8.2e-3 (last position 5.0e-4 / 4.0e-3)
FULL_TOL = {("c1", torch.float16): 2.12e-3, ("c1", torch.bfloat16): 1.65e-2,      # full depth, 27 + 32 layers: 1.69e-3 / 1.32e-2
            ("c2", torch.float16): 1.46e-3, ("c2", torch.bfloat16): 1.43e-2,      #                             1.14e-3 / 1.14e-2
            ("c3", torch.float16): 1.32e-3, ("c3", torch.bfloat16): 1.16e-2}      # C3 = the benchmarked sample:  1.05e-3 (predicted 1.02e-3) / 9.25e-3
# C3 sequence length, 2 + 2 layers: residual stream max over 29 M elements / rel-rms, ViT features, last-position logits
C3LEN_TOL = {torch.float16: (1.13e-3, 9.2e-4, 7.3e-4, 7.1e-4),                    # measured 9.0e-4, 7.4e-4, 5.8e-4, 5.7e-4
             torch.bfloat16: (8.4e-3, 7.3e-3, 5.1e-3, 5.0e-3)}                    # measured 6.8e-3, 5.8e-3, 4.1e-3, 4.0e-3


def err_stats(got, ref):
    d = (got.float() - ref.float())
    return d.abs().max().item(), d.abs().max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


@pytest.fixture(scope="module")
def ops():
    from leopard_amd.ops import Ops
    return Ops()


def build_engine(cfg, ops, dtype):
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.weights import EngineWeights, SynthSource
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, torch.device(DEV), dtype), dtype)
    return LeopardEngine(cfg, W, ops=ops, device=torch.device(DEV))


def sample_inputs(cfg, n_images, w, h, seed=0):
    """Reference-shaped sample: tiler -> u8 tiles + prompt ids (EVAL:384-446 with synthetic images/ids)."""
    from PIL import Image
    from leopard_amd.tiler import tile_sample, to_u8_tiles
    imgs = [Image.fromarray(synth_image_u8(seed + i, w, h)) for i in range(n_images)]
    vit_inputs, plan = tile_sample(imgs)
    u8 = to_u8_tiles(vit_inputs)
    ids = synth_prompt_ids(plan.vit_inputs_per_image, cfg, seed=seed)
    return u8, torch.from_numpy(ids).reshape(1, -1), plan


@pytest.fixture(scope="module")
def mid_oracle():
    """mid config (full width, 2 ViT + 2 LLM layers), one 800x500 image -> thumbnail + 2 tiles."""
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    cfg = mid_config()
    u8, ids, plan = sample_inputs(cfg, 1, 800, 500)
    assert u8.shape[0] == 3
    W = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    logits, parts = O.prefill_logits(ids, torch.from_numpy(siglip_normalize(u8)), W, cfg, return_parts=True)
    gen = O.greedy_generate(ids, torch.from_numpy(siglip_normalize(u8)), W, cfg, max_new_tokens=3)
    return cfg, u8, ids, logits, parts, gen


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_mid_config_prefill_vs_oracle(ops, mid_oracle, dtype):
    cfg, u8, ids, logits, parts, _ = mid_oracle
    eng = build_engine(cfg, ops, dtype)
    res = eng.prefill(ids.to(DEV), torch.from_numpy(u8).to(DEV), all_logits=True, keep_parts=True)
    tol = LOGIT_TOL[dtype]
    v_abs, v_nrm, v_rms = err_stats(res.parts["vit"].cpu().view(3, 676, -1), parts["vit"])
    t_abs, t_nrm, t_rms = err_stats(res.parts["visual_tokens"].cpu().view(3, 169, -1), parts["visual_tokens"])
    e_abs, e_nrm, e_rms = err_stats(res.parts["inputs_embeds"].cpu(), parts["inputs_embeds"][0])
    a_abs, a_nrm, a_rms = err_stats(res.logits_all.cpu(), logits[0])
    l_abs, l_nrm, l_rms = err_stats(res.logits_last.cpu(), logits[0, -1])
    print(f"[mid {dtype}] (max-abs, normalised-max, rel-rms): vit ({v_abs:.2e},{v_nrm:.2e},{v_rms:.2e}) "
          f"visual ({t_abs:.2e},{t_nrm:.2e},{t_rms:.2e}) embeds ({e_abs:.2e},{e_nrm:.2e},{e_rms:.2e}) "
          f"logits_all ({a_abs:.2e},{a_nrm:.2e},{a_rms:.2e}) last ({l_abs:.2e},{l_nrm:.2e},{l_rms:.2e})")
    assert res.seq_len == logits.shape[1]
    assert v_nrm <= tol and t_nrm <= tol and e_nrm <= tol
    assert a_nrm <= tol and l_nrm <= tol
    assert torch.equal(res.logits_all.cpu().argmax(-1)[-1], logits[0, -1].argmax())


def test_mid_config_generate_vs_oracle(ops, mid_oracle):
    cfg, u8, ids, _, _, gen = mid_oracle
    eng = build_engine(cfg, ops, torch.float16)
    out = eng.generate(ids.to(DEV), torch.from_numpy(u8).to(DEV), max_new_tokens=3, eos_token_id=())
    assert out.shape == gen.shape and out.device.type == "cuda"
    assert torch.equal(out.cpu(), gen)


def test_merge_mismatch_raises_before_launch(ops, mid_oracle):
    cfg, u8, ids, *_ = mid_oracle
    eng = build_engine(cfg, ops, torch.float16)
    with pytest.raises(ValueError, match="number of image tokens"):
        eng.prefill(ids.to(DEV), torch.from_numpy(u8[:2]).to(DEV))


# ---- FULL depth (27 + 32 layers) vs committed oracle fixtures ------------------------------------------------------------------------
# tests/golden/{c1,c2,c3}_full_depth.npz are written by tools/gen_fulldepth_fixtures.py: the fp32 oracle and the oracle with the kernels'
# 16-bit hand-over roundings emulated, run ONCE on host cores (C3 = the benchmarked configuration: 140 TFLOP per run).  The tests regenerate
# the inputs from their seeds (checked against the fixture's SHA-256) and compare the HIP path with the stored last-position logits and with
# the stored fp32 residual stream on a few probe rows after every one of the 59 layers.
FULL_CASES = {"c1": (1, 336, 336, 1, 228), "c2": (1, 1344, 896, 7, 1242), "c3": (6, 1344, 896, 42, 7187)}      # images, W, H -> ViT inputs, S
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def probe_rows(name, shape):
    """The rows tools/gen_fulldepth_fixtures.py keeps per layer (ViT stream [N, 676, D] / LLM stream [1, S, D])."""
    if name.startswith("vit"):
        return [(t, r) for t in sorted({0, shape[0] - 1}) for r in (0, 337, 675)]
    S = shape[1]
    return [(0, r) for r in sorted({S // 3, S // 2, max(S - 2, 0), S - 1})]


class FullDepthFixture:
    def __init__(self, case):
        import hashlib
        cfg = full_config()
        n, w, h, n_vit, S = FULL_CASES[case]
        z = np.load(os.path.join(GOLDEN, f"{case}_full_depth.npz"))
        self.u8, self.ids, _ = sample_inputs(cfg, n, w, h)
        assert list(z["meta"]) == [n, w, h, n_vit, S] and self.u8.shape[0] == n_vit
        assert np.array_equal(z["ids"], self.ids.numpy()), "prompt synthesiser drifted from the fixture"
        assert hashlib.sha256(np.ascontiguousarray(self.u8).tobytes()).digest() == z["tiles_sha256"].tobytes(), "tiler output drifted from the fixture"
        self.S, self.n_vit = S, n_vit
        self.ref = torch.from_numpy(z["logits_fp32"])
        self.emu = {k[len("logits_emu_"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("logits_emu_")}
        self.names = [str(x) for x in z["trace_names"]]
        off, wid = z["probe_offsets"], z["probe_width"]
        self.probe = {nm: torch.from_numpy(z["probe_fp32"][off[i]:off[i + 1]]).view(-1, int(wid[i])) for i, nm in enumerate(self.names)}
        self.pred = {k[len("trace_relrms_emu_"):]: dict(zip(self.names, z[k].tolist())) for k in z.files if k.startswith("trace_relrms_emu_")}
        self.oracle_seconds = float(z["oracle_seconds"][0])

    def probe_of(self, name, x):
        """Probe rows of the HIP path's stream `x` ([N * 676, D] or [S, D]) in the fixture's layout."""
        if name.startswith("vit"):
            x = x.view(self.n_vit, -1, x.shape[-1])
            return torch.stack([x[t, r] for t, r in probe_rows(name, x.shape)])
        return torch.stack([x[r] for _, r in probe_rows(name, (1, x.shape[0]))])


@pytest.fixture(scope="module")
def full_depth_oracle():
    done = {}

    class Cases:
        def __getitem__(self, name):
            if name not in done:
                done[name] = FullDepthFixture(name)
            return done[name]
    return Cases()


def rel_rms(a, b):
    return ((a.float() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt()).item()


def run_full_depth(ops, fx, dtype, split=False, precision=None, lo4_vit=None, lo4_rows=None):
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.weights import EngineWeights, SynthSource
    cfg = full_config()
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, torch.device(DEV), dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device=torch.device(DEV))
    eng.precision = precision or ("split" if split else "fast")
    if lo4_vit is not None:
        eng.lo4_vit = lo4_vit                                             # default "auto": the tower corrected for short samples (C1), not for C2 / C3
    if lo4_rows is not None:
        eng.lo4_rows = lo4_rows
    probes = {}
    eng.trace = lambda name, x: probes.__setitem__(name, fx.probe_of(name, x.detach()).float().cpu())
    res = eng.prefill(fx.ids.to(DEV), torch.from_numpy(fx.u8).to(DEV))
    got = res.logits_last.float().cpu()
    assert res.seq_len == fx.S
    del eng, W
    torch.cuda.empty_cache()
    return got, probes


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", ["c1", "c2", "c3"])
def test_full_depth_vs_oracle(ops, full_depth_oracle, case, dtype):
    """BASELINE configs C1, C2 and C3 (the benchmarked 6 x 1344x896 sample, 42 ViT inputs, S = 7187) at FULL depth and width: 27 SigLIP + 32
    Llama-3.1-8B layers, last-position logits and the fp32 residual stream after every layer (probe rows) vs the fp32 oracle
    (EVAL:248-333 semantics), with the figure stated against north_star's 1e-3."""
    fx = full_depth_oracle[case]
    got, probes = run_full_depth(ops, fx, dtype)
    a, n, r = err_stats(got, fx.ref)
    print(f"[{case} full depth {dtype}] vs fp32 oracle ({fx.oracle_seconds:.0f} s of host time, committed): max-abs {a:.3e}  normalised-max {n:.3e} "
          f"(north_star 1e-3: {'met' if n <= 1e-3 else 'x%.2f' % (n / 1e-3)})  rel-rms {r:.3e}  max|logit| {fx.ref.abs().max():.3f}  "
          f"argmax equal = {int(got.argmax()) == int(fx.ref.argmax())}")
    assert n <= FULL_TOL[(case, dtype)]
    assert int(got.argmax()) == int(fx.ref.argmax())
    tag = {torch.float16: "fp16", torch.bfloat16: "bf16"}[dtype]
    # layer by layer on the probe rows: measured error of the fp32 residual stream, and (where the fixture holds the emulating oracle for
    # this type) the predicted budget beside it
    worst = 0.0
    for name in fx.names:
        if name not in probes:
            continue
        m = rel_rms(probes[name], fx.probe[name])
        pr = fx.pred.get(tag, {}).get(name)
        if pr is not None and pr > 0 and name not in ("vit.embed", "llm.embed"):
            worst = max(worst, m / pr)
            assert m <= 2.0 * pr + 1e-6, f"{name}: measured rel-rms {m:.3e} vs predicted {pr:.3e} on the probe rows"
    last = fx.names[-1]
    print(f"[{case} full depth {dtype}] residual stream after {last} (probe rows): measured rel-rms {rel_rms(probes[last], fx.probe[last]):.3e}"
          + (f", predicted {fx.pred[tag][last]:.3e}; worst measured / predicted over the 59 layers {worst:.2f}" if tag in fx.pred else ""))
    if tag in fx.emu:
        a2, n2, r2 = err_stats(got, fx.emu[tag])
        ap, np_, rp = err_stats(fx.emu[tag], fx.ref)
        print(f"[{case} full depth {dtype}] vs rounding-emulating oracle: normalised-max {n2:.3e} rel-rms {r2:.3e};  "
              f"predicted budget (emulated vs fp32 oracle): normalised-max {np_:.3e} rel-rms {rp:.3e}")
        # The emulating oracle is a statistical twin, not a bit-twin: roundings are amplified chaotically through 59 layers, so
        # two runs with the SAME rounding points are as far from each other (sqrt(2) x) as each is from fp32.  What must hold:
        # the measured error IS the predicted 16-bit hand-over budget (profiles/r02_error_growth.txt: equal to <1 % at every
        # layer), and the two 16-bit runs are no further apart than two independent draws of that noise.
        assert 0.7 * rp <= r <= 1.4 * rp, f"measured rel-rms {r:.3e} vs predicted budget {rp:.3e}"
        assert r2 <= 1.8 * rp


@pytest.mark.parametrize("case", ["c1", "c2", "c3"])
def test_full_depth_split_operands_meets_1e_3(ops, full_depth_oracle, case):
    """north_star's figure, literally: with engine.split_operands (every A operand of every layer linear handed over as a hi + lo pair of
    fp16 values, the GEMMs at 2 K) the last-position logits of C1, C2 and C3 at FULL depth are within 1e-3 of the fp32 reference, normalised
    by the logit scale (predicted by the oracle with those hand-overs exact: 5.1e-4 / 2.0e-4, profiles/r03_split_operand_study_c*.txt).
    The mode costs ~1.9x the prefill time (bench.py --split-operands); the production schedule's budget tests are above."""
    fx = full_depth_oracle[case]
    got, _ = run_full_depth(ops, fx, torch.float16, split=True)
    a, n, r = err_stats(got, fx.ref)
    print(f"[{case} full depth fp16, split operands] vs fp32 oracle: max-abs {a:.3e}  normalised-max {n:.3e}  rel-rms {r:.3e}")
    assert n <= 1.0e-3 and int(got.argmax()) == int(fx.ref.argmax())


@pytest.mark.parametrize("case", ["c1", "c2", "c3"])
def test_full_depth_lo4_meets_1e_3(ops, full_depth_oracle, case):
    """north_star's figure on the schedule the bench line is quoted on (round 5): engine.precision = "lo4" — the fast schedule + the fp4 image of
    every LLM layer-linear operand's rounding residual multiplied into the same accumulators (+ 25 % matrix time on those GEMMs, not + 100 %).
    Last-position logits of C1, C2 and C3 at FULL depth within 1e-3 of the fp32 reference, normalised by the logit scale (the absolute figure is
    printed beside it); predicted by the oracle that emulates exactly this arithmetic: C1 7.9e-4, and 6.1e-4 with the SigLIP tower corrected as
    well (profiles/r05_lowbit_correction_study_c1*.txt; the next test)."""
    fx = full_depth_oracle[case]
    got, probes = run_full_depth(ops, fx, torch.float16, precision="lo4")
    a, n, r = err_stats(got, fx.ref)
    last = fx.names[-1]
    print(f"[{case} full depth fp16, lo4 correction] vs fp32 oracle: max-abs {a:.3e}  normalised-max {n:.3e}  rel-rms {r:.3e}  max|logit| {fx.ref.abs().max():.3f}  "
          f"residual stream after {last} (probe rows) rel-rms {rel_rms(probes[last], fx.probe[last]):.3e}  argmax equal = {int(got.argmax()) == int(fx.ref.argmax())}")
    assert n <= 1.0e-3 and int(got.argmax()) == int(fx.ref.argmax())
    # and it is the fast schedule's error that was removed: the fp32 residual stream after the last layer is well inside the 16-bit budget
    # (the SigLIP tower's share of that budget stays: lo4 corrects the LLM layer linears by default) — on the probe rows that carry the
    # correction under the default row policy (engine.lo4_rows = "auto": the trailing rows of a long sequence; the last two probe rows are S - 2, S - 1)
    pred16 = fx.pred.get("fp16", {}).get(last)
    if pred16:
        assert rel_rms(probes[last][-2:], fx.probe[last][-2:]) <= 0.8 * pred16


@pytest.mark.parametrize("rows", ["all", 1, 64, 256, 1024])
@pytest.mark.parametrize("case", ["c2", "c3"])
def test_full_depth_lo4_row_policies(ops, full_depth_oracle, case, rows):
    """WHICH ROWS need the correction (round 6, engine.lo4_rows; predicted by tools/lo4_policy_study.py with the oracle's lo_row_start): the
    last-position logits are dominated by the roundings on the last row's own path; the other rows' roundings are averaged over ~S keys.  'all' is
    round 5's schedule, 256 the first half of round 6; the default 'auto' (the last 16 rows of a sequence longer than 1024) is what test_full_depth_lo4_meets_1e_3 runs."""
    fx = full_depth_oracle[case]
    got, _ = run_full_depth(ops, fx, torch.float16, precision="lo4", lo4_rows=rows)
    a, n, r = err_stats(got, fx.ref)
    print(f"[{case} full depth fp16, lo4 on the last {rows} rows] vs fp32 oracle: max-abs {a:.3e}  normalised-max {n:.3e}  rel-rms {r:.3e}")
    assert int(got.argmax()) == int(fx.ref.argmax())
    if rows in ("all", 256, 1024):
        assert n <= 1.0e-3


@pytest.mark.parametrize("case", ["c1", "c2", "c3"])
def test_full_depth_lo4_with_the_tower_corrected_too(ops, full_depth_oracle, case):
    """engine.lo4_vit = True (LMI_LO4_VIT=1, bench.py --lo4-vit 1): the correction phase on the SigLIP layer linears as well.  Measured: C1 6.1e-4
    (7.3e-4 without), C2 3.3e-4 (3.9e-4), C3 2.37e-4 (2.35e-4: no difference on the benchmarked sample) for + 6 % of the step — which is why the
    default corrects the LLM layers only (predicted for C1: 7.9e-4, profiles/r05_lowbit_correction_study_c1_sites.txt)."""
    fx = full_depth_oracle[case]
    got, _ = run_full_depth(ops, fx, torch.float16, precision="lo4", lo4_vit=True)
    a, n, r = err_stats(got, fx.ref)
    print(f"[{case} full depth fp16, lo4 incl. the SigLIP tower] vs fp32 oracle: max-abs {a:.3e}  normalised-max {n:.3e}  rel-rms {r:.3e}")
    assert n <= 1.0e-3 and int(got.argmax()) == int(fx.ref.argmax())


@pytest.mark.parametrize("case", ["c2"])
def test_full_depth_bf16_precision_modes(ops, full_depth_oracle, case):
    """BASELINE configs[1] names bf16.  With 8 significand bits the attention's own operands (q, k, v, P: untouched by either precision mode) cost
    ~1.6e-3 at this depth, so NO bf16 schedule meets 1e-3; the modes still do what they are for — lo4 removes most of the layer-linear hand-over
    roundings, split all of them — and the figures are printed for DESIGN.md 2.1.  The 1e-3 line of configs[1] is the fp16 engine in lo4 mode."""
    fx = full_depth_oracle[case]
    base, _ = run_full_depth(ops, fx, torch.bfloat16)
    lo4, _ = run_full_depth(ops, fx, torch.bfloat16, precision="lo4")
    split, _ = run_full_depth(ops, fx, torch.bfloat16, precision="split")
    nb, nl, ns = err_stats(base, fx.ref)[1], err_stats(lo4, fx.ref)[1], err_stats(split, fx.ref)[1]
    print(f"[{case} full depth bf16] normalised-max vs fp32 oracle: fast {nb:.3e}  lo4 {nl:.3e}  split {ns:.3e}")
    assert nl < 0.6 * nb and ns < 0.6 * nb
    assert int(lo4.argmax()) == int(fx.ref.argmax()) and int(split.argmax()) == int(fx.ref.argmax())


# ---- fp8 linears (BASELINE configs[4]; leopard_amd.fp8): the error budget of e4m3 operands, predicted and measured --------------------
def _fp8_calibration(cfg):
    u8, ids, _ = sample_inputs(cfg, 1, 700, 420, seed=50)          # a different image and prompt than the evaluated sample
    return [(ids.to(DEV), torch.from_numpy(u8).to(DEV))]


def test_configs4_combination_fp8_packed_batch_graph_encode(ops):
    """BASELINE configs[4] as ONE combination (fp8 linears x several samples packed in one pass x HIP-graph-captured encode), at the mid
    depth: prefill_batch under an fp8 plan with graph_encode on gives, per sample, exactly the logits of that sample's own fp8 prefill
    (kernels are row-independent: packing and graph replay change no bit), the encode really is replayed from a graph, and the batch
    path takes the fused fp8 hand-overs (no conversion / RoPE launches)."""
    cfg = mid_config()
    eng = build_engine(cfg, ops, torch.float16)
    eng.enable_fp8(_fp8_calibration(cfg))
    samples = []
    for n, w, h, seed in [(2, 1344, 896, 21), (1, 800, 500, 22), (3, 700, 700, 23)]:
        u8, ids, _ = sample_inputs(cfg, n, w, h, seed=seed)
        samples.append((ids, torch.from_numpy(u8).to(DEV)))
    singles = [eng.prefill(ids, tiles).logits_last.clone() for ids, tiles in samples]
    calls = []
    for name in ("quantize_fp8", "rope_qk", "attention_fp8out", "rope_qkv_fp8"):
        fn = getattr(ops, name)
        setattr(ops, name, (lambda f, n_: (lambda *a, **k: (calls.append(n_), f(*a, **k))[1]))(fn, name))
    try:
        eng.graph_encode = True
        for rep in range(2):                                        # capture, then replay
            logits, seq_lens = eng.prefill_batch(samples)
            for j, one in enumerate(singles):
                assert torch.equal(logits[j], one), (rep, j)
        assert len(eng._encode_graphs) == 1 and torch.isfinite(logits).all()
        assert "rope_qkv_fp8" in calls and "attention_fp8out" in calls and "quantize_fp8" not in calls and "rope_qk" not in calls
    finally:
        for name in ("quantize_fp8", "rope_qk", "attention_fp8out", "rope_qkv_fp8"):
            delattr(ops, name)


def test_mid_config_fp8_vs_oracle(ops, mid_oracle):
    """2 + 2 layers at full width with fp8 linears: the logits sit where the oracle that rounds the same operands (activations AND
    weights) to e4m3 says they must — fp8 is a reduced-precision line of its own (never the 1e-3 headline), so what is asserted is
    measured == predicted, not a small number."""
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    cfg, u8, ids, logits, parts, _ = mid_oracle
    eng = build_engine(cfg, ops, torch.float16)
    base = eng.prefill(ids.to(DEV), torch.from_numpy(u8).to(DEV), all_logits=True).logits_all.cpu()
    plan = eng.enable_fp8(_fp8_calibration(cfg))
    assert len(plan.vit) == cfg.vision_config.num_hidden_layers and len(plan.llm) == cfg.text_config.num_hidden_layers
    res = eng.prefill(ids.to(DEV), torch.from_numpy(u8).to(DEV), all_logits=True, keep_parts=True)
    W = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    with O.emulate_rounding(torch.float16, operand_dtype=torch.float8_e4m3fn):
        emu, eparts = O.prefill_logits(ids, torch.from_numpy(siglip_normalize(u8)), W, cfg, return_parts=True)
    _, vn, vr = err_stats(res.parts["vit"].cpu().view(3, 676, -1), parts["vit"])
    _, vpn, vpr = err_stats(eparts["vit"], parts["vit"])
    _, n, r = err_stats(res.logits_all.cpu(), logits[0])
    _, pn, pr = err_stats(emu[0], logits[0])
    _, n2, r2 = err_stats(res.logits_all.cpu(), emu[0])
    _, n16, r16 = err_stats(base, logits[0])
    print(f"[mid fp8] ViT features: measured (norm-max {vn:.2e}, rel-rms {vr:.2e}) predicted ({vpn:.2e}, {vpr:.2e});  logits_all: measured "
          f"({n:.2e}, {r:.2e}) predicted ({pn:.2e}, {pr:.2e}) vs emulating oracle ({n2:.2e}, {r2:.2e});  f16 path ({n16:.2e}, {r16:.2e})")
    assert r > 3 * r16                                       # the fp8 schedule ran
    assert 0.7 * vpr <= vr <= 1.4 * vpr and 0.7 * pr <= r <= 1.4 * pr
    assert r2 <= 1.8 * pr
    # the Llama attention arithmetic on the fp8 pipe as well (engine.fp8_attention): predicted by the oracle that also rounds q / k / v / P
    eng.fp8_attention = True
    res8 = eng.prefill(ids.to(DEV), torch.from_numpy(u8).to(DEV), all_logits=True).logits_all.cpu()
    eng.fp8_attention = False
    with O.emulate_rounding(torch.float16, operand_dtype=torch.float8_e4m3fn, fp8_attention=True):
        emu8 = O.prefill_logits(ids, torch.from_numpy(siglip_normalize(u8)), W, cfg)[0]
    _, n8, r8 = err_stats(res8, logits[0])
    _, pn8, pr8 = err_stats(emu8, logits[0])
    _, _, r88 = err_stats(res8, res.logits_all.cpu())
    print(f"[mid fp8 + fp8 attention arithmetic] logits_all: measured ({n8:.2e}, {r8:.2e}) predicted ({pn8:.2e}, {pr8:.2e});  vs the fp8-linears result: rel-rms {r88:.2e}")
    assert 0.7 * pr8 <= r8 <= 1.4 * pr8 and r88 > 0
    # the captured vision encode follows the schedule: graphs taken under fp8 replay fp8, and are dropped when the plan is removed
    eng.graph_encode = True
    t8 = torch.from_numpy(u8).to(DEV)
    for _ in range(2):
        assert torch.equal(eng.prefill(ids.to(DEV), t8, all_logits=True).logits_all, res.logits_all)
    eng.fp8 = None
    for _ in range(2):
        assert torch.equal(eng.prefill(ids.to(DEV), t8, all_logits=True).logits_all.cpu(), base)


def test_full_depth_fp8_c1_vs_oracle(ops, full_depth_oracle):
    """C1 at full depth (27 + 32 layers) with fp8 linears: measured error == the predicted e4m3 budget."""
    cfg = full_config()
    fx = full_depth_oracle["c1"]
    u8, ids, ref, emu = fx.u8, fx.ids, fx.ref, fx.emu
    eng = build_engine(cfg, ops, torch.float16)
    eng.enable_fp8(_fp8_calibration(cfg))
    got = eng.prefill(ids.to(DEV), torch.from_numpy(u8).to(DEV)).logits_last.cpu()
    _, n, r = err_stats(got, ref)
    _, pn, pr = err_stats(emu["fp8"], ref)
    _, n2, r2 = err_stats(got, emu["fp8"])
    print(f"[c1 full depth fp8] measured: normalised-max {n:.3e} rel-rms {r:.3e};  predicted (fp8-emulating oracle vs fp32): {pn:.3e} / {pr:.3e};  "
          f"HIP vs emulating oracle: {n2:.3e} / {r2:.3e};  argmax equal = {int(got.argmax()) == int(ref.argmax())}")
    assert 0.7 * pr <= r <= 1.4 * pr
    assert r2 <= 1.8 * pr
    # the Llama layers' attention arithmetic on the fp8 pipe as well (engine.fp8_attention: e4m3 q / k / v / P): a few more e4m3 roundings
    # per layer on top of the eight operands of its four linears — the error of the line stays of the same size
    eng.fp8_attention = True
    got8 = eng.prefill(ids.to(DEV), torch.from_numpy(u8).to(DEV)).logits_last.cpu()
    _, n8, r8 = err_stats(got8, ref)
    _, _, r88 = err_stats(got8, got)
    print(f"[c1 full depth fp8 + fp8 attention arithmetic] normalised-max {n8:.3e} rel-rms {r8:.3e} vs fp32 (fp8 linears only: {r:.3e});  vs the fp8-linears "
          f"result: rel-rms {r88:.3e};  argmax equal = {int(got8.argmax()) == int(ref.argmax())}")
    assert not torch.isnan(got8).any() and r8 <= 2.0 * pr
    del eng
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def c3_length_oracle():
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    cfg = mid_config()
    u8, ids, plan = sample_inputs(cfg, 6, 1344, 896)
    assert u8.shape[0] == 42
    Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    otrace = []
    with O.emulate_rounding(None, trace=otrace):
        ref, parts = O.prefill_logits(ids, torch.from_numpy(siglip_normalize(u8)), Wt, cfg, last_only=True, return_parts=True)
    return cfg, u8, ids, ref, parts, dict(otrace)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_c3_sequence_length_vs_oracle(ops, c3_length_oracle, dtype):
    """The C3 sample (6 x 1344x896 -> 42 ViT inputs, 7098 visual tokens, S = 7187) through 2 + 2 full-width layers against
    the fp32 oracle: the fp32 residual stream after the last LLM layer on EVERY row (29 GEMM row tiles incl. the 19-row tail,
    softmax over up to 7187 keys), the ViT features of all 42 tiles, and the last-position logits."""
    cfg, u8, ids, ref, parts, otrace = c3_length_oracle
    eng = build_engine(cfg, ops, dtype)
    trace = {}
    eng.trace = lambda name, x: trace.__setitem__(name, x.detach().float().cpu().clone())
    res = eng.prefill(ids.to(DEV), torch.from_numpy(u8).to(DEV), keep_parts=True)
    assert res.seq_len == 7187
    last = f"llm.{cfg.text_config.num_hidden_layers - 1}"
    x_abs, x_nrm, x_rms = err_stats(trace[last], otrace[last][0])
    per_row = (trace[last] - otrace[last][0]).abs().amax(dim=1) / otrace[last][0].abs().max()
    v_abs, v_nrm, v_rms = err_stats(res.parts["vit"].cpu().view(42, 676, -1), parts["vit"])
    l_abs, l_nrm, l_rms = err_stats(res.logits_last.cpu(), ref[0, 0])
    print(f"[C3 length {dtype}] residual stream after the last layer, all 7187 rows: normalised-max {x_nrm:.3e} rel-rms {x_rms:.3e} "
          f"(worst row {int(per_row.argmax())}); ViT features 42 tiles: {v_nrm:.3e} / {v_rms:.3e}; logits: {l_nrm:.3e} / {l_rms:.3e}")
    tx_nrm, tx_rms, t_vit, t_logits = C3LEN_TOL[dtype]
    assert x_nrm <= tx_nrm and x_rms <= tx_rms
    assert v_nrm <= t_vit and l_nrm <= t_logits
    assert int(res.logits_last.argmax()) == int(ref[0, 0].argmax())


def test_c3_size_properties(ops):
    """Full C3 size (6 x 1344x896 -> 42 tiles, S = 7187), full-width layers, reduced depth for run time:
    (1) ViT + projector are tile-permutation equivariant, bit-exactly (tiles are independent sequences);
    (2) packing two samples into one varlen launch gives bit-identical logits to separate launches;
    (3) causality: changing the last prompt token leaves every earlier position's logits bit-identical;
    (4) two identical runs are bit-identical."""
    cfg = mid_config()
    dtype = torch.bfloat16
    eng = build_engine(cfg, ops, dtype)
    u8, ids, plan = sample_inputs(cfg, 6, 1344, 896)
    assert u8.shape[0] == 42 and plan.tiles_per_image == [6] * 6
    tiles = torch.from_numpy(u8).to(DEV)
    vis = eng.encode_images(tiles)
    perm = torch.randperm(42, generator=torch.Generator().manual_seed(0)).to(DEV)
    vis_p = eng.encode_images(tiles[perm].contiguous())
    assert torch.equal(vis.view(42, 169, -1)[perm], vis_p.view(42, 169, -1))
    res = eng.prefill(ids.to(DEV), None, all_logits=True, visual_tokens=vis)
    assert res.seq_len == 7187 and res.n_tiles == 42
    res2 = eng.prefill(ids.to(DEV), None, all_logits=True, visual_tokens=vis)
    assert torch.equal(res.logits_all, res2.logits_all)
    ids2 = ids.clone()
    ids2[0, -1] = (ids2[0, -1] + 1) % 1000
    res3 = eng.prefill(ids2.to(DEV), None, all_logits=True, visual_tokens=vis)
    assert torch.equal(res.logits_all[:-1], res3.logits_all[:-1])
    assert not torch.equal(res.logits_all[-1], res3.logits_all[-1])
    # packed varlen: [sample A | short sample B] in one launch == separate launches
    u8b, idsb, _ = sample_inputs(cfg, 1, 336, 336, seed=9)
    visb = eng.encode_images(torch.from_numpy(u8b).to(DEV))
    xa, xb = eng.embed_merge(ids.to(DEV), vis), eng.embed_merge(idsb.to(DEV), visb)
    la, _ = eng.llm_prefill(xa.clone(), [xa.shape[0]])
    lb, _ = eng.llm_prefill(xb.clone(), [xb.shape[0]])
    lab, _ = eng.llm_prefill(torch.cat([xa, xb]), [xa.shape[0], xb.shape[0]])
    assert torch.equal(lab[0], la[0]) and torch.equal(lab[1], lb[0])


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fast", "lo4", "lo4+vit", "split"])
def test_prefill_batch_equals_per_sample_prefill(ops, precision):
    """BASELINE config C5 shape (a batch of multi-image samples) at the mid depth: one packed pass over three samples with
    different image counts / sizes gives, per sample, exactly the logits of its own prefill call — on the fast schedule and on the
    low-bit-corrected one (the residual images are position-independent too)."""
    cfg = mid_config()
    eng = build_engine(cfg, ops, torch.float16)
    eng.precision = precision.split("+")[0]
    eng.lo4_vit = True if precision.endswith("+vit") else "auto"          # "auto": the short sample's ViT inputs are corrected, the others' are not — in the pack and alone
    shapes = [(1, 800, 500, 3), (2, 1344, 896, 5), (1, 336, 336, 7)]
    samples = []
    for n, w, h, seed in shapes:
        u8, ids, _ = sample_inputs(cfg, n, w, h, seed=seed)
        samples.append((ids, torch.from_numpy(u8).to(DEV)))
    batch_logits, seq_lens = eng.prefill_batch(samples)
    assert len(seq_lens) == 3 and batch_logits.shape[0] == 3
    for i, (ids, tiles) in enumerate(samples):
        one = eng.prefill(ids, tiles)
        assert one.seq_len == seq_lens[i]
        assert torch.equal(one.logits_last, batch_logits[i])


@pytest.mark.gpu
def test_generate_batch_decodes_together_and_equals_per_sample_generate(ops):
    """f4: generate_batch = one packed prefill + ONE captured decode step per token for the whole batch (pooled KV cache, skinny-M
    projections).  Per sample: the first new token is bit-identical by construction (same prefill); the continuation is compared with
    the batch-1 generate() — the projections sum in a different order (MFMA tiles vs FMA chains), so a token may only differ where the
    batch-1 logits' top-2 gap is below the 16-bit noise; the test asserts equality and, where it fails, that it was such a near tie."""
    from leopard_amd.engine import KVCache
    cfg = mid_config()
    eng = build_engine(cfg, ops, torch.float16)
    shapes = [(1, 800, 500, 3), (2, 1344, 896, 5), (1, 336, 336, 7), (1, 364, 364, 9)]
    samples = []
    for n, w, h, seed in shapes:
        u8, ids, _ = sample_inputs(cfg, n, w, h, seed=seed)
        samples.append((ids, torch.from_numpy(u8).to(DEV)))
    T = 6
    singles = [eng.generate(ids, tiles, max_new_tokens=T, eos_token_id=()) for ids, tiles in samples]
    steps = []
    body = eng._batch_decode_body
    eng._batch_decode_body = lambda st: (steps.append(st.B), body(st))[1]
    batch = eng.generate_batch(samples, max_new_tokens=T, eos_token_id=())
    eng._batch_decode_body = body
    assert set(steps) == {4} and len(steps) == 2                       # warm-up + capture: afterwards the graph replays (no Python body)
    st = eng._batch_states[4]
    assert st.graph is not None
    for (ids, tiles), one, got in zip(samples, singles, batch):
        S_in = ids.shape[1]
        assert got.shape == one.shape and int(got[0, S_in]) == int(one[0, S_in])
        if not torch.equal(one, got):
            j = int((one[0] != got[0]).nonzero()[0])
            cache = KVCache(cfg, one.shape[1] + 256 * 8, torch.float16, DEV)
            res = eng.prefill(ids, tiles, cache=cache)
            lg = None
            nxt = int(one[0, S_in])
            for t in range(S_in + 1, j + 1):
                lg = eng.decode_step(nxt, cache).clone()
                nxt = int(one[0, t])
            top2 = lg.topk(2).values
            assert float(top2[0] - top2[1]) <= 2e-3 * float(lg.abs().max()), (j, top2)
    again = eng.generate_batch(samples, max_new_tokens=T, eos_token_id=())
    assert all(torch.equal(a, b) for a, b in zip(batch, again)) and eng._batch_states[4] is st


@pytest.mark.gpu
def test_generate_stream_continuous_batching_on_device(ops):
    """f4 continuous batching on the device (LeopardEngine.generate_stream; mid configuration = full width, 2 + 2 layers): 14 samples of
    mixed image counts / sizes through 4 decode slots with ONE captured step — slots are retired by the device-side stop rule (eos ids,
    token budget) and re-admitted between replays; outputs == per-sample generate() (a token may differ only on a near tie of the batch-1
    logits, as for generate_batch), no step is run by the Python body after the capture, and the live-slot occupancy is reported."""
    from leopard_amd.engine import KVCache
    cfg = mid_config()
    eng = build_engine(cfg, ops, torch.float16)
    shapes = [(1, 800, 500), (2, 1344, 896), (1, 336, 336), (1, 364, 364), (0, 0, 0), (1, 700, 420), (3, 500, 500), (1, 336, 336),
              (0, 0, 0), (2, 800, 500), (1, 1344, 896), (1, 364, 364), (0, 0, 0), (1, 500, 800)]
    samples = []
    for i, (n, w, h) in enumerate(shapes):
        if n == 0:
            ids = torch.from_numpy(np.random.default_rng(200 + i).integers(3, 7000, (1, 9 + 5 * i)))
            samples.append((ids, None))
        else:
            u8, ids, _ = sample_inputs(cfg, n, w, h, seed=100 + i)
            samples.append((ids, torch.from_numpy(u8).to(DEV)))
    T = 12
    free = [eng.generate(ids, tiles, max_new_tokens=T, eos_token_id=()) for ids, tiles in samples]
    # eos ids that make some samples stop early: sample 2's 4th new token, sample 7's 2nd, sample 9's FIRST (finished by its prefill)
    eos = tuple({int(free[2][0, samples[2][0].shape[1] + 3]), int(free[7][0, samples[7][0].shape[1] + 1]), int(free[9][0, samples[9][0].shape[1]])})
    singles = [eng.generate(ids, tiles, max_new_tokens=T, eos_token_id=eos) for ids, tiles in samples]
    assert min(o.shape[1] - s[0].shape[1] for o, s in zip(singles, samples)) == 1
    bodies = []
    body = eng._batch_decode_body
    eng._batch_decode_body = lambda st: (bodies.append(st.B), body(st))[1]
    stats = {}
    got = eng.generate_stream(samples, batch_size=4, max_new_tokens=T, eos_token_id=eos, stats=stats)
    eng._batch_decode_body = body
    assert len(bodies) == 2 and eng._batch_states[4].graph is not None            # warm-up + capture; every step afterwards is a replay
    n_diff = 0
    for (ids, tiles), one, out in zip(samples, singles, got):
        S_in = ids.shape[1]
        if torch.equal(one, out):
            continue
        n_diff += 1                                                    # allowed only as a near tie of the batch-1 logits at the first difference
        j = int((one[0, :min(one.shape[1], out.shape[1])] != out[0, :min(one.shape[1], out.shape[1])]).nonzero()[0])
        cache = KVCache(cfg, one.shape[1] + 2048, torch.float16, DEV)
        eng.prefill(ids, tiles, cache=cache)
        lg, nxt = None, int(one[0, S_in])
        for t in range(S_in + 1, j + 1):
            lg = eng.decode_step(nxt, cache).clone()
            nxt = int(one[0, t])
        top2 = lg.topk(2).values
        assert float(top2[0] - top2[1]) <= 2e-3 * float(lg.abs().max()), (j, top2)
    occ = stats["live_slot_steps"] / stats["slot_steps"]
    print(f"[generate_stream] 14 samples, 4 slots, T = {T}: {stats['steps']} steps, live-slot occupancy {occ:.2f}, "
          f"{n_diff} sample(s) differ from generate() on a near tie")
    assert n_diff <= 2 and occ > 0.5
    again = eng.generate_stream(samples, batch_size=4, max_new_tokens=T, eos_token_id=eos)
    assert all(torch.equal(a, b) for a, b in zip(got, again))
    eng.release_batch_state()


@pytest.mark.gpu
def test_c5_size_batch_properties(ops):
    """BASELINE config C5 size (8 samples x 8 images of 1344x896 -> 320 ViT inputs, 8 x 6861 tokens in one packed pass) at the
    mid depth: every sample of the packed batch reproduces its own single-sample prefill bit for bit, and equal samples give
    equal logits wherever they sit in the batch."""
    cfg = mid_config()
    eng = build_engine(cfg, ops, torch.float16)
    u8, ids, plan = sample_inputs(cfg, 8, 1344, 896, seed=11)
    assert plan.n_vit_inputs == 40                                   # 8 x (thumbnail + 4 tiles)
    tiles = torch.from_numpy(u8).to(DEV)
    u8b, idsb, _ = sample_inputs(cfg, 8, 1344, 896, seed=12)
    tiles_b = torch.from_numpy(u8b).to(DEV)
    samples = [(ids, tiles), (idsb, tiles_b)] * 4                    # 8 samples, 320 ViT inputs
    logits, seq_lens = eng.prefill_batch(samples)
    assert len(seq_lens) == 8 and all(s == seq_lens[0] for s in seq_lens) and seq_lens[0] == ids.shape[1] + 40 * 168
    assert torch.isfinite(logits).all()
    for i in range(2, 8):
        assert torch.equal(logits[i], logits[i % 2])
    one = eng.prefill(ids, tiles)
    assert torch.equal(one.logits_last, logits[0])


@pytest.mark.gpu
def test_configs4_fp8_at_the_8x8_size_with_graph_encode(ops):
    """BASELINE configs[4] AS WRITTEN — batch 8 x 8 images of 1344x896 (320 ViT inputs, 8 x 6861 tokens in one packed pass), fp8 (e4m3)
    layer linears, HIP-graph-captured encode — at the mid depth, in property form: every sample of the packed fp8 batch reproduces its
    own single-sample fp8 prefill bit for bit (packing and graph replay change no bit), equal samples give equal logits wherever they
    sit, the encode really is replayed from a graph, and the fp8 schedule really ran (its logits differ from the f16 schedule's by the
    e4m3 budget, not by zero)."""
    cfg = mid_config()
    eng = build_engine(cfg, ops, torch.float16)
    u8, ids, plan = sample_inputs(cfg, 8, 1344, 896, seed=11)
    assert plan.n_vit_inputs == 40
    tiles = torch.from_numpy(u8).to(DEV)
    u8b, idsb, _ = sample_inputs(cfg, 8, 1344, 896, seed=12)
    tiles_b = torch.from_numpy(u8b).to(DEV)
    f16_one = eng.prefill(ids, tiles).logits_last.clone()
    eng.enable_fp8(_fp8_calibration(cfg))
    eng.graph_encode = True
    one_a = eng.prefill(ids, tiles).logits_last.clone()
    one_b = eng.prefill(idsb, tiles_b).logits_last.clone()
    samples = [(ids, tiles), (idsb, tiles_b)] * 4                    # 8 samples, 320 ViT inputs
    for rep in range(2):                                             # capture of the 320-input encode, then its replay
        logits, seq_lens = eng.prefill_batch(samples)
        assert len(seq_lens) == 8 and seq_lens[0] == ids.shape[1] + 40 * 168 and torch.isfinite(logits).all()
        for i in range(8):
            assert torch.equal(logits[i], one_a if i % 2 == 0 else one_b), (rep, i)
    assert any(k[0] == 320 for k in eng._encode_graphs)
    rel = ((one_a - f16_one).pow(2).mean().sqrt() / f16_one.pow(2).mean().sqrt()).item()
    print(f"[configs[4] size, mid depth] fp8 vs f16 schedule, relative RMS of the last-position logits: {rel:.3e}")
    assert 1e-2 < rel < 0.5
    # the configuration bench.py times for configs[4]: the Llama attention arithmetic on the fp8 pipe as well (per-sequence tile images of the
    # packed batch) — the same property: every sample of the packed batch == its own prefill, bit for bit
    eng.fp8_attention = True
    a8_a, a8_b = eng.prefill(ids, tiles).logits_last.clone(), eng.prefill(idsb, tiles_b).logits_last.clone()
    logits8, _ = eng.prefill_batch(samples)
    for i in range(8):
        assert torch.equal(logits8[i], a8_a if i % 2 == 0 else a8_b), i
    rel8 = ((a8_a - one_a).pow(2).mean().sqrt() / one_a.pow(2).mean().sqrt()).item()
    print(f"[configs[4] size, mid depth] + fp8 attention arithmetic: relative RMS vs the fp8-linears logits {rel8:.3e}")
    assert torch.isfinite(logits8).all() and 0 < rel8 < 0.5


def test_graph_captured_encode_is_bit_identical(ops):
    """BASELINE config 5's "hipGraph-captured encode": vision tower + projector replayed from a HIP graph per ViT-input count
    == the eager launches, bit for bit, for two different counts and fresh pixel data on every replay."""
    cfg = mid_config()
    eng = build_engine(cfg, ops, torch.float16)
    g = torch.Generator().manual_seed(5)
    for n in (3, 7, 3):
        tiles = torch.randint(0, 256, (n, 364, 364, 3), generator=g, dtype=torch.uint8).to(DEV)
        eng.graph_encode = False
        want = eng.encode_images(tiles).clone()
        eng.graph_encode = True
        got = eng.encode_images(tiles)
        assert torch.equal(got, want)
    assert sorted(k[0] for k in eng._encode_graphs) == [3, 7]            # one graph per (ViT-input count, launch stream)
    u8, ids, _ = sample_inputs(cfg, 1, 800, 500)
    a = eng.prefill(ids.to(DEV), torch.from_numpy(u8).to(DEV)).logits_last.clone()
    eng.graph_encode = False
    b = eng.prefill(ids.to(DEV), torch.from_numpy(u8).to(DEV)).logits_last
    assert torch.equal(a, b)


@pytest.mark.parametrize("case", ["text_only", "twenty_thumbnails", "one_token", "mixed_sizes"])
def test_edge_case_samples_vs_oracle(ops, case):
    """Edge cases of the reference's sample shapes on the device, vs the fp32 oracle (mid configuration):
    text only (no image tokens, no ViT work); 20 small images (tile budget 50 - 20 spread thin: every image rounds to one
    natural tile -> thumbnails only, N = 20, EVAL:26-58); a one-token prompt; one sample mixing a 336x336, a portrait and a
    panorama image (ragged tile counts per image)."""
    from PIL import Image
    from leopard_amd.tiler import siglip_normalize, tile_sample, to_u8_tiles
    from oracle import leopard_oracle as O
    cfg = mid_config()
    eng = build_engine(cfg, ops, torch.float16)
    W = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    rng = np.random.default_rng(12)
    if case in ("text_only", "one_token"):
        ids = torch.from_numpy(rng.integers(0, 7000, (1, 1 if case == "one_token" else 37)))
        res = eng.prefill(ids.to(DEV), None)
        emb = torch.nn.functional.embedding(ids.reshape(-1), W["language_model.model.embed_tokens.weight"]).unsqueeze(0)
        ref = O.llama_forward(emb, torch.arange(ids.shape[1]).unsqueeze(0), W, cfg, last_only=True)[0, 0]
        assert res.seq_len == ids.shape[1] and res.n_tiles == 0
    else:
        sizes = [(300 + 7 * i, 280 + 5 * i) for i in range(20)] if case == "twenty_thumbnails" else [(336, 336), (500, 1400), (1900, 420)]
        imgs = [Image.fromarray(synth_image_u8(60 + i, w, h)) for i, (w, h) in enumerate(sizes)]
        vit_inputs, plan = tile_sample(imgs)
        if case == "twenty_thumbnails":
            assert plan.tiles_per_image == [0] * 20 and plan.n_vit_inputs == 20
        else:
            assert plan.tiles_per_image[0] == 0 and len(set(plan.tiles_per_image)) > 1
        u8 = to_u8_tiles(vit_inputs)
        ids = torch.from_numpy(synth_prompt_ids(plan.vit_inputs_per_image, cfg, seed=3)).reshape(1, -1)
        res = eng.prefill(ids.to(DEV), torch.from_numpy(u8).to(DEV))
        ref = O.prefill_logits(ids, torch.from_numpy(siglip_normalize(u8)), W, cfg, last_only=True)[0, 0]
        assert res.n_tiles == plan.n_vit_inputs
    a, n, r = err_stats(res.logits_last.cpu(), ref)
    print(f"[edge {case}] S={res.seq_len} normalised-max {n:.3e} rel-rms {r:.3e}")
    assert n <= LOGIT_TOL[torch.float16] and int(res.logits_last.argmax()) == int(ref.argmax())


def test_full_size_engine_holds_one_copy_of_the_llm_weights(ops):
    """VERDICT r03 item 6 at the full Llama-3.1-8B size: the default engine keeps the layer linears ONCE (packed order, 13.96 GB) — prefill,
    batch-1 decode and batched decode read the same tensors; a prefill gives the bits of an engine that keeps the nn.Linear layout,
    greedy decoding the same tokens, and after batched generation the resident weights (all of them, with the packed head copy of the
    batched step) stay under 19 GB."""
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.weights import EngineWeights, SynthSource, is_packed
    cfg = full_config()
    dev, dtype = torch.device(DEV), torch.float16
    mk = lambda: EngineWeights.build(cfg, SynthSource(cfg, ops, dev, dtype), dtype)
    W0, W1 = mk(), mk()
    plain = LeopardEngine(cfg, W0, ops=ops, device=dev, pack_llm_weights=False)
    eng = LeopardEngine(cfg, W1, ops=ops, device=dev)
    assert eng.llm_packed and not plain.llm_packed
    nbytes = lambda t: 0 if t is None else t.numel() * t.element_size()
    layers = lambda W: sum(nbytes(t) for L in W.llm_layers for t in (L.qkv_w, L.qkv_w_rope, L.o_w, L.gu_w, L.down_w))
    assert layers(W1) == 32 * 2 * (6144 * 4096 + 4096 * 4096 + 28672 * 4096 + 4096 * 14336) and layers(W0) > layers(W1)
    ids = torch.from_numpy(synth_prompt_ids([1], cfg, seed=3)).reshape(1, -1)
    tiles = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (1, 364, 364, 3), dtype=np.uint8)).to(DEV)
    a, b = plain.prefill(ids, tiles), eng.prefill(ids, tiles)
    assert torch.equal(a.logits_last, b.logits_last)
    t0, t1 = plain.generate(ids, tiles, max_new_tokens=6, eos_token_id=()), eng.generate(ids, tiles, max_new_tokens=6, eos_token_id=())
    same = int((t0 == t1).sum()) == t0.numel()
    print(f"[one weights copy] prefill logits bit-identical; 6 greedy tokens {'equal' if same else 'differ (near tie)'}: {t1[0, -6:].tolist()}")
    del plain, W0
    torch.cuda.empty_cache()
    samples = [(ids, tiles), (torch.from_numpy(synth_prompt_ids([], cfg, seed=4)).reshape(1, -1), None), (ids, tiles)]
    outs = eng.generate_stream(samples, batch_size=2, max_new_tokens=5, eos_token_id=())
    assert torch.equal(outs[0], outs[2]) and outs[0].shape[1] == ids.shape[1] + 5
    assert getattr(eng, "_skinny_pack", None) is None and is_packed(eng._head_pack)
    resident = layers(W1) + sum(nbytes(t) for t in (W1.embed, W1.lm_head, eng._head_pack))
    resident += sum(nbytes(t) for L in W1.vit_layers for t in vars(L).values() if torch.is_tensor(t))
    print(f"[one weights copy] resident weights {resident / 1e9:.2f} GB (layers {layers(W1) / 1e9:.2f}), torch allocator {torch.cuda.memory_allocated() / 1e9:.2f} GB "
          f"with the KV pools and workspaces")
    assert resident < 19e9
    eng.release_batch_state()


def test_prefills_on_two_streams_do_not_share_scratch(ops):
    """the engine's activation workspaces were keyed by stage only, so two prefills in flight on different HIP streams
    (bench.py --inflight 2) wrote the same scratch.  Workspaces are per (stage, launch stream) now: two samples prefilled concurrently on two
    streams, three times over, give bit for bit the logits of the same samples prefilled one after the other — in the fast and lo4 schedules."""
    cfg = mid_config()
    eng = build_engine(cfg, ops, torch.float16)
    samples = []
    for n, w, h, seed in [(2, 1344, 896, 31), (1, 800, 500, 32)]:
        u8, ids, _ = sample_inputs(cfg, n, w, h, seed=seed)
        samples.append((ids, torch.from_numpy(u8).to(DEV)))
    streams = [torch.cuda.Stream(device=DEV) for _ in samples]
    for mode in ("fast", "lo4"):
        eng.precision = mode
        ref = [eng.prefill(ids, tiles).logits_last.clone() for ids, tiles in samples]
        torch.cuda.synchronize()
        for _ in range(3):
            outs = []
            for (ids, tiles), st in zip(samples, streams):
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    outs.append(eng.prefill(ids, tiles).logits_last)
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(outs, ref)), mode
    assert len({k[1] for k in eng._workspaces}) >= 2            # one workspace per launch stream

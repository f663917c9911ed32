#!/usr/bin/env python3
"""bench.py — Leopard-LLaVA multi-image prefill throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank/GPU)

A "step" is one pass of the hot path over one synthetic sample of BASELINE config C3 — 6 images of 1344x896 ->
adaptive tiler (on the GPU) -> 42 ViT inputs (364x364) -> SigLIP-SO400M (27 layers) -> pixel-shuffle + projector -> 7098
visual tokens merged into a 7187-token sequence -> Llama-3.1-8B prefill (32 layers, KV cache written) ->
last-position logits ("tiler -> logits of last position").  The source pixels (u8 HWC) and the token ids
are resident in HBM / on the host when the timed region starts; weights are seeded synthetic values of the real
architecture (no checkpoints exist offline).  `ms_per_step_excl_tiler` times the same step from ready-made tiles.

Multi-GPU (`--gpus N`, one rank per GPU): the headline shards by SAMPLE exactly as the reference does
(run_eval_llava_siglip_multiimg.sh:9-11, one process per GPU over dataset shards, no collective on the data path): every
rank prefills its own sample, per-GPU work is fixed ("weak" scaling), value = ranks x images / max-over-ranks time.  The
same line also carries `"tp"`: ONE sample per step on all ranks (tile-sharded ViT + all-gather, sequence-parallel
tensor-parallel LLM over RCCL; strong scaling), measured after the headline in one child process per rank (a crash or hang
there costs the `tp` object, not the line); `--parallelism tp` makes it the headline.  `backend`, `rccl_ranks` and `comm_bytes_per_step` say what carried the ranks.

One JSON line on rank 0:  metric/value/unit + roofline (dominant kernel = the MFMA GEMM family incl. its fused norm / RoPE
epilogues, HIP-event timed on the launch stream; `dominant` = the single largest shape) + cpu_baseline (the CPU oracle's
end-to-end C1 prefill on the host cores, plus the C3 figure extrapolated from a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from leopard_amd.config import full_config  # noqa: E402
from leopard_amd.synth import synth_image_u8, synth_prompt_ids  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0     # dense bf16/fp16 peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_FP8_TFLOPS = 5000.0 # dense fp8 peak (same guide); only the --dtype fp8 line is priced against it
MFMA_PEAK_FP4_TFLOPS = 10000.0  # dense fp4 / fp6 peak (same guide): the correction phase of the lo4 schedule
HBM_PEAK_GBS = 8000.0


def algorithmic_flops(cfg, n_tiles: int, S: int) -> dict:
    """SURVEY.md 8(d) per-unit figures (2*MAC): ViT tile, projector tile, LLM prefill of S tokens, last-token head."""
    v, t = cfg.vision_config, cfg.text_config
    T, d, ff = v.num_patches, v.hidden_size, v.intermediate_size
    vit_tile = 2 * T * v.patch_dim * d + v.num_hidden_layers * (2 * T * (4 * d * d + 2 * d * ff) + 4 * T * T * d)
    proj_tile = 2 * cfg.tokens_per_tile * (cfg.projector_in * t.hidden_size + t.hidden_size ** 2)
    qkv = t.hidden_size * (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
    lin = t.num_hidden_layers * 2 * S * (qkv + t.num_attention_heads * t.head_dim * t.hidden_size + 3 * t.hidden_size * t.intermediate_size)
    attn = t.num_hidden_layers * 2 * t.num_attention_heads * t.head_dim * S * (S + 1)
    head = 2 * t.hidden_size * t.vocab_size
    return {"vit": n_tiles * vit_tile, "projector": n_tiles * proj_tile, "llm_linear": lin, "llm_attention": attn,
            "lm_head_last": head, "total": n_tiles * (vit_tile + proj_tile) + lin + attn + head}


def vit_attention_flops(cfg) -> int:
    """QK^T + PV of one ViT input through all SigLIP layers (part of algorithmic_flops()["vit"]; not a layer linear: no correction phase)."""
    v = cfg.vision_config
    return v.num_hidden_layers * 4 * v.num_patches * v.num_patches * v.hidden_size


def make_sample(cfg, n_images: int, width: int, height: int, seed: int):
    """Host side of the path (EVAL:384-446): synthetic u8 images -> tiler (PIL) -> u8 ViT inputs + prompt ids."""
    from PIL import Image
    from leopard_amd.tiler import tile_sample, to_u8_tiles
    imgs = [Image.fromarray(synth_image_u8(seed * 100 + i, width, height)) for i in range(n_images)]
    t0 = time.perf_counter()
    vit_inputs, plan = tile_sample(imgs)
    u8 = to_u8_tiles(vit_inputs)
    host_s = time.perf_counter() - t0
    ids = synth_prompt_ids(plan.vit_inputs_per_image, cfg, seed=seed)
    return u8, ids, plan, host_s, [np.asarray(im) for im in imgs]


class GemmTimer:
    """HIP-event pairs around every lmi_gemm launch, recorded on the launch stream."""

    def __init__(self):
        self.records = []
        self.bytes = 0              # algorithmic operand + output bytes of the timed launches
        self.lo4_flops = 0.0        # algorithmic FLOPs of the launches that also ran the fp4 correction phase
        self.sel_frac = {}

    def wrap(self, ops):
        inner = ops.gemm
        timer = self

        def gemm(a, w, out, *args, **kw):
            M = kw.get("M", None)
            if M is None:
                M = a.shape[0]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            r = inner(a, w, out, *args, **kw)
            e1.record(torch.cuda.current_stream())
            timer.records.append((2.0 * M * w.shape[0] * w.shape[1], e0, e1, (M, w.shape[0], w.shape[1])))
            timer.bytes += (M + w.shape[0]) * w.shape[1] * w.element_size() + out.numel() * out.element_size()
            return r
        ops.gemm = gemm
        # the folded-norm / RoPE variants are GEMMs of the same family: lmi_gemm_ex(a, w, out, ...), lmi_rmsnorm_rope(a, w, qkv, ...)
        self._inner_ex, self._inner_rope = ops.gemm_ex, ops.rmsnorm_rope

        def timed(fn):
            def call(a, w, out, *args, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream())
                r = fn(a, w, out, *args, **kw)
                e1.record(torch.cuda.current_stream())
                timer.records.append((2.0 * a.shape[0] * w.shape[0] * w.shape[1], e0, e1, (a.shape[0], w.shape[0], w.shape[1])))
                timer.bytes += (a.shape[0] + w.shape[0]) * w.shape[1] * w.element_size() + out.numel() * out.element_size()
                return r
            return call
        ops.gemm_ex, ops.rmsnorm_rope = timed(ops.gemm_ex), timed(ops.rmsnorm_rope)
        # precision "lo4": the same family with the fp4 correction phase (a = Lo4Act: 16-bit rows + fp4 image + block scales).  FLOPs booked
        # are the ALGORITHMIC 2 M N K of the linear — the correction phase's extra matrix work is overhead, not useful work
        self._inner_lo4, self._inner_rope_lo4 = ops.gemm_lo4, ops.rmsnorm_rope_lo4

        def timed_lo4(fn):
            def call(a, w, w4, out, *args, **kw):
                M = a.hi.shape[0]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream())
                r = fn(a, w, w4, out, *args, **kw)
                e1.record(torch.cuda.current_stream())
                timer.records.append((2.0 * M * w.shape[0] * w.shape[1], e0, e1, (M, w.shape[0], w.shape[1])))
                # row selection (engine.lo4_rows): only the 256-row tiles that hold a selected row run the fp4 k-tiles and read the images
                cf = 1.0
                if a.unit_sel is not None:
                    cf = timer.sel_frac.get(a.unit_sel.data_ptr())
                    if cf is None:                                                       # (one device read per selection table, not per launch)
                        u = a.unit_sel.cpu().numpy().astype(bool)
                        u = np.pad(u, (0, (-len(u)) % 4)).reshape(-1, 4).any(axis=1)      # 256-row tiles = 4 units
                        cf = timer.sel_frac[a.unit_sel.data_ptr()] = min(1.0, float(u.sum()) * 256 / M)
                timer.lo4_flops += 2.0 * M * w.shape[0] * w.shape[1] * cf
                timer.bytes += ((M + w.shape[0]) * w.shape[1] * w.element_size() + out.numel() * out.element_size()
                                + int((a.img.numel() + a.sc.numel()) * cf) + w4.img.numel())
                return r
            return call
        ops.gemm_lo4, ops.rmsnorm_rope_lo4 = timed_lo4(ops.gemm_lo4), timed_lo4(ops.rmsnorm_rope_lo4)
        # --dtype fp8: the fp8 GEMMs are their own family (own records, priced against the fp8 peak)
        self._inner_fp8 = ops.gemm_fp8
        self.fp8_records, self.fp8_bytes = [], 0

        def gemm_fp8(a, w, out, *args, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            r = timer._inner_fp8(a, w, out, *args, **kw)
            e1.record(torch.cuda.current_stream())
            timer.fp8_records.append((2.0 * a.shape[0] * w.shape[0] * w.shape[1], e0, e1, (a.shape[0], w.shape[0], w.shape[1])))
            timer.fp8_bytes += (a.shape[0] + w.shape[0]) * w.shape[1] + out.numel() * out.element_size()
            return r
        ops.gemm_fp8 = gemm_fp8
        return inner

    def unwrap(self, ops, inner):
        ops.gemm, ops.gemm_ex, ops.rmsnorm_rope, ops.gemm_fp8 = inner, self._inner_ex, self._inner_rope, self._inner_fp8
        ops.gemm_lo4, ops.rmsnorm_rope_lo4 = self._inner_lo4, self._inner_rope_lo4

    def use_fp8_family(self):
        """Make the fp8 launches the family summary() / dominant() describe; returns (flops, ms, n) of the 16-bit launches left."""
        rest = self.summary()
        self.records, self.bytes = self.fp8_records, self.fp8_bytes
        return rest

    def times(self):
        """Elapsed ms of every record.  The passes launch the same sequence: a launch's time is the MEDIAN over the passes (an event pair also
        spans any moment the host fell behind the device between the two records; the median of three drops one such stall without booking
        every launch at its best case)."""
        torch.cuda.synchronize()
        t = [r[1].elapsed_time(r[2]) for r in self.records]
        p = getattr(self, "passes", 1)
        if p > 1 and len(t) % p == 0 and len(t) > 0:
            n = len(t) // p
            if all(self.records[i][3] == self.records[i + k * n][3] for k in range(1, p) for i in range(n)):
                mid = [sorted(t[i + k * n] for k in range(p))[(p - 1) // 2] for i in range(n)]
                t = mid * p
        return t

    def summary(self):
        flops = sum(r[0] for r in self.records)
        return flops, sum(self.times()), len(self.records)

    def dominant(self, peak=MFMA_PEAK_TFLOPS):
        """The (M, N, K) shape with the largest total time: launches, average ms, TFLOP/s."""
        by = {}
        for (fl, e0, e1, shape), ms in zip(self.records, self.times()):
            t = by.setdefault(shape, [0, 0.0, fl])
            t[0] += 1
            t[1] += ms
        shape, (n, ms, fl) = max(by.items(), key=lambda kv: kv[1][1])
        return {"shape_MNK": list(shape), "launches": n, "avg_launch_ms": round(ms / n, 4), "achieved": round(fl / (ms / n * 1e-3) / 1e12, 1),
                "frac": round(fl / (ms / n * 1e-3) / 1e12 / peak, 4)}


def kernel_source_hash() -> str:
    """sha256 over the kernel sources: stamps PMC summaries so that a stale one is refused (tools/collect_traffic.sh)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(REPO, "leopard_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def roofline_from_timer(ops, run_once, passes: int, fp8: bool):
    """HIP-event timing of every GEMM-family launch of `passes` runs of `run_once` on the launch stream -> the `roofline` object
    of the bench line (bound = MFMA; the family is priced against the dense peak of ITS operand type)."""
    timer = GemmTimer()
    timer.passes = passes
    inner = timer.wrap(ops)
    torch.cuda.synchronize()
    for _ in range(passes):
        run_once()
    peak, family, rest = MFMA_PEAK_TFLOPS, "lmi::gemm_kernel / gemm_stagger_kernel (all epilogues, incl. the fused RMSNorm / RoPE / KV-append ones)", None
    if fp8:
        rest = timer.use_fp8_family()
        peak, family = MFMA_PEAK_FP8_TFLOPS, "lmi::gemm_kernel / gemm_stagger_kernel, fp8 e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4), all epilogues"
    gflops, gms, n = timer.summary()
    dominant = timer.dominant(peak)
    timer.unwrap(ops, inner)
    achieved = (gflops / n) / (gms / n * 1e-3) / 1e12
    r = {"bound": "mfma", "kernel": family, "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
         "traffic": None, "traffic_unit": "HBM-side bytes per launch (PMC)", "algorithmic_bytes_per_launch": round(timer.bytes / n),
         "kernel_source_hash": kernel_source_hash(), "dominant": dominant, "launches_per_step": n // passes,
         "avg_launch_ms": round(gms / n, 4), "gemm_ms_per_step": round(gms / passes, 2)}
    if timer.lo4_flops > 0 and not fp8:
        r["correction_phase_flops_share"] = round(timer.lo4_flops / gflops, 4)
    if rest is not None and rest[2] > 0:
        r["f16_gemms_left"] = {"launches_per_step": rest[2] // passes, "ms_per_step": round(rest[1] / passes, 2),
                               "achieved": round(rest[0] / (rest[1] * 1e-3) / 1e12, 1), "peak": MFMA_PEAK_TFLOPS}
    return r


def cpu_baseline_c1(cfg, ops, dev):
    """the CPU oracle's END-TO-END prefill of config C1 (1 x 336x336 image + 32-token question, S = 228,
    27 SigLIP + 32 Llama-3.1-8B layers, fp32, same synthetic parameters) on the host cores: 1 warm-up (the bounded sample of
    cpu_baseline_sample below) + 3 timed runs, median.  Returns (seconds, algorithmic TFLOP of C1)."""
    from PIL import Image
    from leopard_amd.tiler import siglip_normalize, tile_sample, to_u8_tiles
    from leopard_amd.weights import SynthSource
    from oracle import leopard_oracle as O
    src = SynthSource(cfg, ops, dev, torch.float16)      # lmi_fill_synthetic == the numpy generator, bit for bit (tests)
    Wt = {name: src.get(name).float().cpu() for name in src.specs}
    vit_inputs, plan = tile_sample([Image.fromarray(synth_image_u8(0, 336, 336))])
    u8 = to_u8_tiles(vit_inputs)
    ids = torch.from_numpy(synth_prompt_ids(plan.vit_inputs_per_image, cfg, seed=0)).reshape(1, -1)
    pix = torch.from_numpy(siglip_normalize(u8))
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        logits = O.prefill_logits(ids, pix, Wt, cfg, last_only=True)
        times.append(time.perf_counter() - t0)
    assert torch.isfinite(logits).all()
    S = ids.shape[1] + u8.shape[0] * (cfg.tokens_per_tile - 1)
    return sorted(times)[1], algorithmic_flops(cfg, u8.shape[0], S)["total"] / 1e12, S


def cpu_baseline_sample(cfg):
    """The CPU oracle (fp32 PyTorch restatement of the reference path) timed on the host cores on a bounded sample
    of the C3 workload: 2 SigLIP layers on 2 tiles + 1 Llama layer at S=1024, converted to images/s of the C3
    workload through the algorithmic FLOP count (C3 itself would take ~7 minutes per sample on the host)."""
    from leopard_amd.synth import param_specs, synth_array
    from oracle import leopard_oracle as O
    specs = {n: (s, k) for n, s, k in param_specs(cfg)}
    want = [n for n in specs if n.startswith("vision_tower.vision_model.encoder.layers.0.")
            or n.startswith("vision_tower.vision_model.encoder.layers.1.") or n.startswith("language_model.model.layers.0.")]
    W = O.weights_from_numpy({n: synth_array(n, *specs[n]) for n in want})
    g = torch.Generator().manual_seed(0)
    xv = torch.randn(2, cfg.vision_config.num_patches, cfg.vision_config.hidden_size, generator=g)
    S = 1024
    xl = torch.randn(1, S, cfg.text_config.hidden_size, generator=g)
    tc = cfg.text_config
    cos, sin = O.rope_tables(torch.arange(S), tc.head_dim, tc.rope_theta, tc.rope_scaling)
    v = cfg.vision_config
    T, d, ff = v.num_patches, v.hidden_size, v.intermediate_size
    fl_v = 2 * 2 * (2 * T * (4 * d * d + 2 * d * ff) + 4 * T * T * d)
    qkv = tc.hidden_size * (tc.num_attention_heads + 2 * tc.num_key_value_heads) * tc.head_dim
    fl_l = 2 * S * (qkv + tc.hidden_size ** 2 + 3 * tc.hidden_size * tc.intermediate_size) + 2 * tc.num_attention_heads * tc.head_dim * S * (S + 1)

    def run():
        with torch.no_grad():
            y = O.siglip_layer(O.siglip_layer(xv, W, 0, cfg), W, 1, cfg)
            z = O.llama_layer(xl, W, 0, cfg, cos, sin)
        return y, z
    run()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    t = sorted(times)[1]
    tflops = (fl_v + fl_l) / t / 1e12
    return tflops, t


def idefics2_flops(cfg, n_img: int, P: int, S: int) -> float:
    """Algorithmic FLOPs (2 x MACs) of a Leopard-Idefics2 prefill: n_img images of P patches each, S text + latent tokens."""
    v, t, pc = cfg.vision_config, cfg.text_config, cfg.perceiver_config
    d, ff, Dt, L = v.hidden_size, v.intermediate_size, t.hidden_size, pc.n_latents
    vit = 2 * P * v.patch_dim * d + v.num_hidden_layers * (2 * P * (4 * d * d + 2 * d * ff) + 4 * P * P * d)
    mp = 2 * P * (2 * d * t.intermediate_size + t.intermediate_size * Dt)
    qd, kd = pc.n_heads * pc.head_dim, pc.num_key_value_heads * pc.head_dim
    perc = pc.depth * (2 * L * Dt * qd + 2 * (P + L) * Dt * 2 * kd + 4 * L * (P + L) * qd + 2 * L * qd * Dt + 2 * L * 3 * Dt * 4 * Dt)
    qkv = Dt * (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
    llm = t.num_hidden_layers * (2 * S * (qkv + Dt * Dt + 3 * Dt * t.intermediate_size) + 2 * t.num_attention_heads * t.head_dim * S * (S + 1)) + 2 * Dt * t.vocab_size
    return float(n_img * (vit + mp + perc) + llm)


def cpu_baseline_idefics2():
    """The Idefics2 CPU oracle (oracle/idefics2_oracle.py) timed on the host cores on a bounded sample: ONE 1344x896 image (-> 980x653,
    3220 patches) + 38 text tokens through full-width layers at reduced depth (2 NaViT + 2 perceiver + 2 Mistral layers, 8k vocabulary);
    median of 3 after 1 warm-up -> host TFLOP/s, which the caller scales to the C4 sample by algorithmic FLOPs."""
    from PIL import Image
    from leopard_amd.config import idefics2_mid_config
    from leopard_amd.synth import idefics2_state_dict_numpy
    from oracle import idefics2_oracle as IO
    cfg = idefics2_mid_config()
    Wt = IO.weights_from_numpy(idefics2_state_dict_numpy(cfg))
    pix = [IO.image_processor(Image.fromarray(synth_image_u8(0, 1344, 896)), cfg.longest_edge)]
    L = cfg.perceiver_config.n_latents
    rng = np.random.default_rng(0)
    ids = torch.tensor([rng.integers(3, 7000, 6).tolist() + [cfg.image_token_id] * L + rng.integers(3, 7000, 32).tolist()])
    P = (pix[0].shape[1] // cfg.vision_config.patch_size) * (pix[0].shape[2] // cfg.vision_config.patch_size)
    IO.prefill_logits(ids, pix, Wt, cfg, last_only=True)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        IO.prefill_logits(ids, pix, Wt, cfg, last_only=True)
        times.append(time.perf_counter() - t0)
    t = sorted(times)[1]
    return idefics2_flops(cfg, 1, P, ids.shape[1]) / t / 1e12, t


FP8_DETAIL = ("fp8 e4m3fn operands (static power-of-two scales, fp32 accumulate) for the qkv / out / fc1 / fc2 linears of the 27 SigLIP "
              "layers and the qkv / o / gate-up / down linears of the 32 Llama layers; f16 attention, patch embed, projector; fp32 "
              "residual stream, norms and head")
FP8_ATTN_DETAIL = ("; Llama attention arithmetic on the fp8 pipe as well: e4m3 q / k / v (static scales) and e4m3 P, QK^T and PV on "
                   "v_mfma_scale_f32_32x32x64_f8f6f4, fp32 softmax statistics and accumulators — an OPT-IN of the fp8 schedule that this bench switches on "
                   "(engine.fp8_attention / LMI_FP8_ATTENTION=1; the library default and --fp8-attention 0 keep the f16 attention arithmetic: "
                   "81.3 vs 76.5 ms per C3 step, profiles/r04_bench_c3_fp8_f16_attention.json)")


FP8_ACCURACY_NOTE = ("STRESS TEST of the fp8 matrix pipe, not a usable precision mode: per-tensor static e4m3 operands through 59 layers of the synthetic "
                     "model leave 0.35 relative RMS on the logits — 10 of 32 greedy tokens agree with the fp32 oracle on the C1 decision fixture "
                     "(tests/test_gpu_decisions.py, profiles/r05_decisions.txt) against 32 of 32 for the fast and lo4 schedules")


def fp8_detail(args):
    return FP8_DETAIL.replace("f16 attention, patch embed", "f16 SigLIP attention, patch embed") + FP8_ATTN_DETAIL if getattr(args, "fp8_attention", 0) else FP8_DETAIL


def enable_fp8(eng, cfg, args):
    """Static activation scales from a 16-bit prefill of a DIFFERENT synthetic sample (other images, other prompt)."""
    u8, ids_np, _, _, _ = make_sample(cfg, 2, args.width, args.height, seed=977)
    eng.enable_fp8([(torch.from_numpy(ids_np).reshape(1, -1), torch.from_numpy(u8).to(eng.device))])
    if getattr(args, "fp8_attention", None) is None:
        args.fp8_attention = int(eng.fp8_attention)                    # the library default (LMI_FP8_ATTENTION)
    eng.fp8_attention = bool(args.fp8_attention)                       # the calibration above recorded the q / k / v ranges either way
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def fixture_parity(args, ctx, res):
    """The TIMED step's last-position logits against the committed full-depth oracle fixture of the same sample (tests/golden/c{1,2,3}_full_depth.npz,
    written by tools/gen_fulldepth_fixtures.py: the fp32 CPU oracle = the reference's arithmetic, EVAL:248-333, on host cores) — the parity figure of
    the benchmarked configuration, at the benchmarked depth, from the benchmarked run.  Data only: no oracle code runs here.  None when the sample is
    not one of the fixtures' (other image arguments, rank > 0, several samples in flight) or for a dtype line without a parity claim (fp8)."""
    import hashlib
    label = config_label(args).lower()
    path = os.path.join(REPO, "tests", "golden", f"{label}_full_depth.npz")
    if res is None or label not in ("c1", "c2", "c3") or not os.path.exists(path) or args.dtype == "fp8":
        return None
    z = np.load(path)
    same_ids = z["ids"].shape == tuple(ctx.ids.shape) and bool((z["ids"] == ctx.ids.cpu().numpy()).all())
    same_tiles = hashlib.sha256(np.ascontiguousarray(ctx.tiles.cpu().numpy()).tobytes()).digest() == z["tiles_sha256"].tobytes()
    if not (same_ids and same_tiles):
        return None
    ref = torch.from_numpy(z["logits_fp32"])
    got = res.logits_last.float().cpu()
    d = got - ref
    tag = {"f16": "fp16", "bf16": "bf16"}.get(args.dtype)
    pred = None
    if tag and f"logits_emu_{tag}" in z.files:
        pred = float((torch.from_numpy(z[f"logits_emu_{tag}"]) - ref).abs().max() / ref.abs().max())
    return {"fixture": os.path.relpath(path, REPO), "vs": "fp32 CPU oracle, full depth (27 + 32 layers), last-position logits",
            "max_abs": round(float(d.abs().max()), 6), "normalised_max": round(float(d.abs().max() / ref.abs().max()), 6),
            "rel_rms": round(float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), 6), "max_abs_logit": round(float(ref.abs().max()), 4),
            "argmax_equal": bool(int(got.argmax()) == int(ref.argmax())),
            "predicted_normalised_max_16bit_operands": None if pred is None else round(pred, 6),
            "north_star_1e-3": "met" if float(d.abs().max() / ref.abs().max()) <= 1e-3 else "x%.2f" % (float(d.abs().max() / ref.abs().max()) / 1e-3)}


def config_label(args) -> str:
    """BASELINE.json configuration the image arguments correspond to (C3 = the metric's own, the default)."""
    key = (args.images, args.width, args.height)
    return {(6, 1344, 896): "C3", (1, 1344, 896): "C2", (1, 336, 336): "C1"}.get(key, "custom")


def metric_name(args) -> str:
    return f"multi-image prefill images/sec (Leopard-LLaVA, {args.images}x{args.width}x{args.height} per sample)"


def bench_c5(args, dev, dtype, rank, world, D):
    """BASELINE config 5 shape: a batch of 8 samples x 8 images of 1344x896 (40 ViT inputs and 6861 tokens per sample), all 8
    samples in ONE packed pass (LeopardEngine.prefill_batch).  --dtype fp8 = the configuration as BASELINE.json words it (e4m3
    operands for the ViT / LLM layer linears); --graph-encode replays the vision encode from a HIP graph (prefill_batch ->
    encode_images)."""
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.gpu_tiler import GpuTiler
    from leopard_amd.ops import Ops
    from leopard_amd.weights import EngineWeights, SynthSource
    cfg = full_config()
    ops = Ops()
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, dev, dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device=dev)
    eng.graph_encode = args.graph_encode
    tiler = GpuTiler(ops, dev)
    n_samples, n_img = 8, 8
    samples = []
    for j in range(n_samples):
        raw = [synth_image_u8((rank * 16 + j) * 100 + i, 1344, 896) for i in range(n_img)]
        tiles, plan = tiler.tile_sample(raw)
        ids = torch.from_numpy(synth_prompt_ids(plan.vit_inputs_per_image, cfg, seed=rank * 16 + j)).reshape(1, -1)
        samples.append((ids, tiles))
    n_tiles = sum(t.shape[0] for _, t in samples)
    if args.dtype == "fp8":
        enable_fp8(eng, cfg, args)

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        eng.prefill_batch(samples)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits, seq_lens = eng.prefill_batch(samples)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    assert torch.isfinite(logits).all()
    fl = sum(algorithmic_flops(cfg, t.shape[0], S)["total"] for (_, t), S in zip(samples, seq_lens))
    out = {"metric": "multi-image prefill images/sec (Leopard-LLaVA, batch 8 x 8 x 1344x896)",
           "value": round(world * n_samples * n_img * args.steps / elapsed, 3), "unit": "images/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           **({"dtype_detail": fp8_detail(args), "accuracy_note": FP8_ACCURACY_NOTE} if args.dtype == "fp8" else {}),
           "config": {"workload": f"C5 shape: {n_samples} samples x {n_img} x (1344x896) -> {n_tiles} ViT inputs, "
                                  f"{sum(seq_lens)} tokens packed in one varlen pass; SigLIP-SO400M + Llama-3.1-8B prefill to "
                                  "last-token logits; synthetic seeded weights", "parallelism": f"sample-sharded x{world}"},
           "graph_encode": bool(args.graph_encode), "encode_graphs_captured": len(eng._encode_graphs),
           "algorithmic_tflop_per_step": round(fl / 1e12, 2),
           "prefill_mfma_frac": round(fl / 1e12 / (elapsed / args.steps) / MFMA_PEAK_TFLOPS, 4),
           "prefill_mfma_frac_note": "algorithmic FLOPs / time against the 2.5 PF dense 16-bit peak" +
                                     (" (mixed-precision step: the fp8 linears' own roofline is below)" if args.dtype == "fp8" else "")}
    if rank == 0 and not args.no_roofline:
        def eager_pass():                       # launches replayed from the captured encode graph cannot be bracketed by events: time the eager form
            keep, eng.graph_encode = eng.graph_encode, False
            try:
                eng.prefill_batch(samples)
            finally:
                eng.graph_encode = keep
        out["roofline"] = roofline_from_timer(ops, eager_pass, 1, args.dtype == "fp8")
        out["roofline"]["traffic_source"] = "not collected for this workload"
    if rank == 0 and world == 1 and args.no_cpu_baseline and args.cpu_tflops > 0:
        out["cpu_baseline"] = extrapolated_cpu_baseline(args.cpu_tflops, n_samples * n_img, fl)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        tflops, t = cpu_baseline_sample(cfg)
        out["cpu_baseline"] = {"value": round(n_samples * n_img / (fl / 1e12 / tflops), 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"CPU oracle (oracle/leopard_oracle.py): 2 SigLIP layers x 2 tiles + 1 Llama layer at S=1024 ({t:.2f} s, "
                                         f"{tflops:.3f} TFLOP/s on {torch.get_num_threads()} host threads), scaled to the batch ({fl / 1e12:.0f} TFLOP) by algorithmic FLOPs"}
    if rank == 0:
        print(json.dumps(out), flush=True)


def bench_idefics2(args, dev, dtype, rank, world, D):
    """BASELINE config 4: Leopard-Idefics2, 4 images of 1344x896 (-> 980x653, 3220 patches, 64 visual tokens each)
    interleaved in a ~312-token prompt; NaViT SigLIP (27L) + modality projection + perceiver (3L) + Mistral-7B (32L)."""
    from PIL import Image
    from leopard_amd.config import idefics2_full_config
    from leopard_amd.engine import KVCache
    from leopard_amd.idefics2 import Idefics2Engine, Idefics2SynthSource, Idefics2Weights, preprocess_image_u8
    from leopard_amd.ops import Ops
    cfg = idefics2_full_config()
    ops = Ops()
    # --parallelism tp (BASELINE configs[3]: "TP=8 LLM over xGMI"): ONE sample per step on all ranks — images round-robin over the ranks
    # for the NaViT tower / perceiver + one all-gather, Mistral tensor-parallel with sequence-parallel norms; every rank gets the same sample
    tp = args.parallelism == "tp" and world > 1
    W = Idefics2Weights.build(cfg, Idefics2SynthSource(cfg, ops, dev, dtype), dtype, tp_rank=rank if tp else 0, tp_size=world if tp else 1)
    eng = Idefics2Engine(cfg, W, ops=ops, device=dev)
    # the reference loads this model in fp16 (idefics2_multiimg.py:27-28): the fast fp16 schedule is its own arithmetic; --precision lo4 = tower + Mistral corrected
    idef_precision = args.precision if getattr(args, "precision", None) in ("fast", "lo4") else "fast"
    eng.precision = idef_precision
    n_img = 4
    seed = 0 if tp else rank
    imgs = [torch.from_numpy(preprocess_image_u8(Image.fromarray(synth_image_u8(seed * 16 + i, 1344, 896)), cfg.longest_edge).copy()).to(dev)
            for i in range(n_img)]
    L = cfg.perceiver_config.n_latents
    rng = np.random.default_rng(seed)
    ids = []
    for _ in range(n_img):
        ids += rng.integers(3, 32000, 6).tolist() + [cfg.image_token_id] * L
    ids += rng.integers(3, 32000, 32).tolist()
    ids = torch.tensor([ids])
    S = ids.shape[1]
    cache = KVCache(cfg, eng.tp_padded_len(S) if tp else S, dtype, dev, tp_size=world if tp else 1)

    def step():
        cache.length = 0
        return eng.prefill(ids, imgs, cache=cache)

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    assert torch.isfinite(res.logits_last).all()
    total = idefics2_flops(cfg, n_img, 3220, S)
    out = {"metric": "multi-image prefill images/sec (Leopard-Idefics2, 4x1344x896 per sample)",
           "value": round((1 if tp else world) * n_img * args.steps / elapsed, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "strong" if tp else "weak",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "precision_mode": PRECISION_NOTE[idef_precision],
           "config": {"workload": f"C4: 4x(1344x896) -> 980x653, 3220 patches each, 64 visual tokens each, S={S}; NaViT SigLIP (27L) + "
                                  "perceiver (3L) + Mistral-7B (32L) prefill to last-token logits; synthetic seeded weights",
                      "parallelism": (f"one sample on {world} ranks: images round-robin + 1 all-gather; TP{world} Mistral with sequence-parallel norms"
                                      if tp else f"sample-sharded x{world}")},
           "algorithmic_tflop_per_step": round(total / 1e12, 2),
           "prefill_mfma_frac": round(total / 1e12 / (elapsed / args.steps) / (MFMA_PEAK_TFLOPS * (world if tp else 1)), 4)}
    if tp:
        out.update({"backend": eng.comm.backend, "rccl_ranks": eng.comm.ranks_seen(), "comm_bytes_per_step": int(eng.comm.sent_bytes / (args.steps + args.warmup)) * world})
    if rank == 0 and seed == 0 and args.dtype == "f16":
        out["parity"] = idefics2_fixture_parity(ids, imgs, res)
    if rank == 0 and not tp and not args.no_roofline:
        out["roofline"] = roofline_from_timer(ops, step, 1, False)
        out["roofline"]["traffic_source"] = "not collected for this workload"
    if rank == 0 and world == 1 and args.no_cpu_baseline and args.cpu_tflops > 0:
        out["cpu_baseline"] = extrapolated_cpu_baseline(args.cpu_tflops, n_img, total)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        tflops, t = cpu_baseline_idefics2()
        out["cpu_baseline"] = {"value": round(n_img / (total / 1e12 / tflops), 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"Idefics2 CPU oracle (fp32 PyTorch restatement, oracle/idefics2_oracle.py) on one 1344x896 image + 38 tokens through "
                                         f"2 + 2 + 2 full-width layers: {t:.2f} s = {tflops:.3f} TFLOP/s on {torch.get_num_threads()} host threads, scaled to "
                                         f"the C4 sample ({total / 1e12:.1f} TFLOP) by algorithmic FLOPs"}
    if rank == 0:
        print(json.dumps(out), flush=True)


def idefics2_fixture_parity(ids, imgs, res):
    """The TIMED step's last-position logits against tests/golden/c4_idefics2_full_depth.npz (tools/gen_idefics2_fixture.py: the fp32 CPU oracle of
    Leopard-Idefics2 at full depth on this very sample; data only).  None when the inputs are not the fixture's."""
    import hashlib
    path = os.path.join(REPO, "tests", "golden", "c4_idefics2_full_depth.npz")
    if not os.path.exists(path):
        return None
    z = np.load(path)
    sha = hashlib.sha256(b"".join(np.ascontiguousarray(a.cpu().numpy()).tobytes() for a in imgs)).digest()
    if not np.array_equal(z["ids"], ids.numpy()) or sha != z["images_sha256"].tobytes():
        return None
    ref, emu = torch.from_numpy(z["logits_fp32"]), torch.from_numpy(z["logits_emu_fp16"])
    got = res.logits_last.float().cpu().reshape(-1)
    scale = ref.abs().max().item()
    d = (got - ref).abs()
    n = d.max().item() / scale
    return {"fixture": "tests/golden/c4_idefics2_full_depth.npz", "vs": "fp32 CPU oracle (oracle/idefics2_oracle.py), full depth (27 + 3 + 32 layers), last-position logits",
            "max_abs": round(d.max().item(), 6), "normalised_max": round(n, 6), "rel_rms": round((d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), 6),
            "max_abs_logit": round(scale, 4), "argmax_equal": int(got.argmax()) == int(ref.argmax()),
            "predicted_normalised_max_16bit_operands": round(((emu - ref).abs().max() / scale).item(), 6),
            "north_star_1e-3": "met" if n <= 1e-3 else "x%.2f" % (n / 1e-3)}


def measure_tp(args, cfg, ops, dev, dtype, rank, world, D, gpu_tiler):
    """ONE sample per step on all ranks (north_star's partitioning): tile-sharded vision encode + one all-gather,
    sequence-parallel tensor-parallel LLM (all-gather / reduce-scatter per half layer over RCCL, LeopardEngine._llm_layers_tp),
    column-parallel last-token head.  Same timed region as the headline (tiler -> last-position logits); strong scaling."""
    from leopard_amd.engine import KVCache, LeopardEngine
    from leopard_amd.weights import EngineWeights, SynthSource
    Wt = EngineWeights.build(cfg, SynthSource(cfg, ops, dev, dtype), dtype, tp_rank=rank, tp_size=world)
    eng = LeopardEngine(cfg, Wt, ops=ops, device=dev)
    eng.fuse_norm_rope = not args.no_fuse
    tp_precision = args.precision if getattr(args, "precision", None) in ("fast", "lo4") else ("lo4" if args.dtype == "f16" and getattr(args, "precision", None) is None else "fast")
    if tp_precision == "lo4" and not eng.lo4_supported():
        tp_precision = "fast"
    eng.precision = tp_precision                 # lo4 (default, f16): the Lo4 pairs travel through the sequence-parallel norms and the all-gathers
    u8, ids_np, plan, _, raw = make_sample(cfg, args.images, args.width, args.height, seed=0)       # the SAME sample on every rank
    raw_dev = [torch.from_numpy(np.ascontiguousarray(r)).to(dev) for r in raw]
    ids = torch.from_numpy(ids_np).reshape(1, -1)
    S = ids.shape[1] + u8.shape[0] * (cfg.tokens_per_tile - 1)
    cache = KVCache(cfg, eng.tp_padded_len(S), dtype, dev, tp_size=world)

    def step():
        cache.length = 0
        return eng.prefill(ids, gpu_tiler.tile_sample(raw_dev)[0], cache=cache)

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    sent0 = eng.comm.sent_bytes
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    assert res.seq_len == S and torch.isfinite(res.logits_last).all()
    fl = algorithmic_flops(cfg, u8.shape[0], S)

    class _Ctx:                                                  # the sample is the fixture's (seed 0): the tensor-parallel logits against the fp32 oracle
        pass
    pc = _Ctx()
    pc.ids, pc.tiles = ids, torch.from_numpy(u8)
    return {"value": round(args.images * args.steps / elapsed, 3), "unit": "images/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "parity": fixture_parity(args, pc, res) if rank == 0 else None,
            "scaling": "strong", "n_gpus": world, "precision_mode": PRECISION_NOTE[tp_precision],
            "parallelism": f"one sample on {world} ranks: ViT inputs sharded {u8.shape[0]} -> {world} + 1 all-gather; TP{world} LLM with "
                           f"sequence-parallel norms ({eng.tp_chunks} row chunks, all-gather + reduce-scatter per half layer in "
                           f"{'fp32' if eng.tp_comm_dtype == torch.float32 else args.dtype}), column-parallel head",
            "backend": eng.comm.backend, "rccl_ranks": eng.comm.ranks_seen(),
            "comm_strict": os.environ.get("LMI_COMM_STRICT") == "1",
            "comm_bytes_per_step_per_rank": int((eng.comm.sent_bytes - sent0) / args.steps),
            "prefill_mfma_frac_of_n_gpus": round(fl["total"] / 1e12 / (elapsed / args.steps) / (MFMA_PEAK_TFLOPS * world), 4)}


DEFAULT_PRECISION = "lo4"     # the schedule whose full-depth logits are within north_star's 1e-3 of the fp32 reference
PRECISION_NOTE = {
    "fast": "fast: one rounding of every activation to the 16-bit compute type per MFMA-operand hand-over",
    "lo4": "lo4: fast + the MX fp4 image of the LLM layer-linear operands' rounding residuals multiplied with an fp4 weight image into the same "
           "accumulators (v_mfma_scale_f32_32x32x64_f8f6f4, + 25 % matrix time on the row tiles that run it), on the ROWS WHOSE LOGITS ARE READ: the "
           "trailing rows of each sequence (engine.lo4_rows = 'auto': every row of a sequence up to 1024 rows, the last 16 rows of a longer one; "
           "LMI_LO4_ROWS=all restores every row) — a row's logits are dominated by the hand-over roundings on its own path through the layers, the "
           "other rows' reach it through the softmax average over ~S keys (see the `lo4_rows` object; algorithmic FLOPs below are the model's, not "
           "the extra MFMA work; --lo4-vit 1 / LMI_LO4_VIT=1 extends the correction to the SigLIP layer linears)",
    "split": "split operands: every A operand of every ViT / LLM layer linear handed over as hi + lo 16-bit values, GEMMs at 2 K "
             "(algorithmic FLOPs below are the model's, not the doubled MFMA work)"}


def extrapolated_cpu_baseline(cpu_tflops: float, units: float, total_flops: float) -> dict:
    """cpu_baseline of an other_configs child run: the host TFLOP/s the PARENT run measured on its bounded CPU-oracle sample (fp32 PyTorch
    restatement of the reference path; cpu_baseline_sample), scaled to this workload by algorithmic FLOPs."""
    return {"value": round(units / (total_flops / 1e12 / cpu_tflops), 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"CPU oracle bounded sample of the parent bench run ({cpu_tflops:.3f} TFLOP/s on {torch.get_num_threads()} host threads), scaled to this "
                      f"workload ({total_flops / 1e12:.1f} TFLOP per step) by algorithmic FLOPs"}


OTHER_CONFIGS = [   # (key, BASELINE.json configuration, extra argv) — short runs appended to the default line
    ("c1_f16_lo4", "configs[0]: 1 x 336x336 + 32-token prompt (the reference's CPU-runnable case)", ["--images", "1", "--width", "336", "--height", "336", "--steps", "10", "--warmup", "3"]),
    ("c2_f16_lo4", "configs[1] shape (1 x 1344x896), fp16 + lo4: the parity-qualified line", ["--images", "1", "--steps", "10", "--warmup", "3"]),
    ("c2_bf16_fast", "configs[1] as worded (1 x 1344x896, bf16)", ["--images", "1", "--dtype", "bf16", "--steps", "10", "--warmup", "3"]),
    ("c3_f16_split", "configs[2] sample, split-operand precision mode (hi + lo 16-bit pairs at 2 K)", ["--precision", "split", "--steps", "3", "--warmup", "1"]),
    ("c4_idefics2", "configs[3]: Leopard-Idefics2, 4 x 1344x896 (one rank; TP in the N > 1 runs), fp16 fast = the reference's own arithmetic for this model",
     ["--workload", "idefics2-c4", "--steps", "10", "--warmup", "3"]),
    ("c4_idefics2_lo4", "configs[3] with the lo4 schedule on the NaViT tower and the Mistral layers (full-depth logits 5.4e-4 of the fp32 oracle's scale)",
     ["--workload", "idefics2-c4", "--precision", "lo4", "--steps", "10", "--warmup", "3"]),
    ("configs4_fp8_graph", "configs[4]: batch 8 x 8 images, fp8 MFMA ViT + LLM prefill, HIP-graph-captured encode",
     ["--workload", "llava-c5", "--dtype", "fp8", "--fp8-attention", "1", "--graph-encode", "--steps", "3", "--warmup", "2"]),
]


def run_other_configs(cpu_tflops: float, timeout_s: float = 240.0) -> dict:
    """The other BASELINE configurations, each as a short child run of this script on the same GPU (own process: own library options, own
    allocator; a failure costs its entry, not the headline).  Every entry keeps the child's own line fields that matter here."""
    import subprocess
    keep = ("value", "unit", "ms_per_step", "steps", "dtype", "precision_mode", "prefill_mfma_frac", "matrix_pipe_frac", "parity", "cpu_baseline",
            "graph_encode", "dtype_detail", "accuracy_note", "lo4_rows")
    res = {}
    for key, what, argv in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--no-other-configs", "--no-fast-line", "--no-cpu-baseline", "--cpu-tflops", f"{cpu_tflops:.4f}"] + argv
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                res[key] = {"config": what, "error": (p.stderr or "no output")[-300:]}
                continue
            d = json.loads(line[-1])
            e = {"config": what, "workload": d.get("config", {}).get("workload"), **{k: d[k] for k in keep if k in d}}
            if "roofline" in d:
                e["roofline"] = {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_incl_correction_phase") if k in d["roofline"]}
            e["wall_s"] = round(time.perf_counter() - t0, 1)
            res[key] = e
        except subprocess.TimeoutExpired:
            res[key] = {"config": what, "error": f"no result within {timeout_s:.0f} s"}
    return res


def run_tp_child(args, rank, world, D, dev) -> dict:
    """The one-sample-on-all-ranks measurement (`--parallelism tp` of this script) as one child process per rank: same RANK / LOCAL_RANK /
    WORLD_SIZE, a rendezvous port of its own (picked by rank 0, agreed over the parent group), the parent's GPU released first.  The
    parent waits at most --tp-timeout seconds and kills exactly the child it started; rank 0 returns the child's "tp" object."""
    import socket
    import subprocess
    port = 0
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    port = int(D.max_over_ranks(float(port), dev))
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}      # the child is its own rendezvous, not the agent's
    env["MASTER_PORT"] = str(port)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--parallelism", "tp", "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--dtype", args.dtype, "--images", str(args.images), "--width", str(args.width), "--height", str(args.height)]
    if args.precision in ("fast", "lo4"):
        cmd += ["--precision", args.precision]
    if args.no_fuse:
        cmd.append("--no-fuse")
    for kv in args.opt:
        cmd += ["--opt", kv]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        so, se = p.communicate(timeout=args.tp_timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        so, se = p.communicate()
        return {"error": f"tensor-parallel measurement (child process of rank {rank}) did not finish within {args.tp_timeout:.0f} s", "stderr_tail": (se or "")[-300:]}
    if rank != 0:
        return {}
    line = [l for l in so.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not line:
        return {"error": f"child process of rank 0 exited with {p.returncode}", "stderr_tail": (se or "no output")[-300:]}
    return json.loads(line[-1]).get("tp", {"error": "child line has no tp object"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16", "fp8"],
                    help="f16 (default, the BASELINE metric's precision) / bf16; fp8 = BASELINE configs[4]: e4m3 operands for the ViT / "
                         "LLM layer linears over an f16 engine (llava-c3 / llava-c5 only) — a separate line, never the headline")
    ap.add_argument("--images", type=int, default=6)
    ap.add_argument("--width", type=int, default=1344)
    ap.add_argument("--height", type=int, default=896)
    ap.add_argument("--workload", default="llava-c3", choices=["llava-c3", "idefics2-c4", "llava-c5"],
                    help="llava-c3 = the BASELINE metric's configuration (default); idefics2-c4 = Leopard-Idefics2, 4 x 1344x896")
    ap.add_argument("--inflight", type=int, default=1,
                    help="independent samples in flight per GPU, each on its own HIP stream (1 = the reference's one-sample-at-a-time loop)")
    ap.add_argument("--parallelism", default="sample", choices=["sample", "tp"],
                    help="N > 1: 'sample' = one sample per rank, no data-path collective (default, weak scaling); 'tp' = ONE sample "
                         "per step on all ranks as the headline (strong scaling).  With 'sample' and N > 1 the one-sample-on-all-ranks "
                         "figure is measured as well and reported under \"tp\" in the same JSON line")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (lmi_set_option), e.g. gemm.wide=7 for A/B runs")
    ap.add_argument("--no-fuse", action="store_true", help="A/B: separate RMSNorm / RoPE launches instead of the fused GEMM epilogues")
    ap.add_argument("--fp8-attention", type=int, default=None, choices=[0, 1],
                    help="--dtype fp8: 1 = the Llama layers' QK^T and PV on the fp8 matrix pipe too (lmi_attn_prep_fp8 + lmi_attn_fp8_fwd), 0 = the f16 "
                         "attention; default = the library's (engine.fp8_attention: LMI_FP8_ATTENTION, off) — the configs[4] entry of the default line "
                         "passes 1 explicitly")
    ap.add_argument("--graph-encode", action="store_true", help="capture the vision encode (ViT + projector) in a HIP graph per ViT-input count")
    ap.add_argument("--split-operands", action="store_true",
                    help="precision mode (NOT the headline): hi + lo split A operands for every layer linear, GEMMs at 2 K — full-depth logits within 1e-3 of fp32")
    ap.add_argument("--precision", default=None, choices=["fast", "lo4", "split"],
                    help="schedule of the 16-bit engines (DESIGN.md 2.1): fast = one rounding per operand hand-over; lo4 = + the fp4 correction "
                         "phase (logits within north_star's 1e-3 at full depth); split = hi + lo 16-bit operand pairs at 2 K")
    ap.add_argument("--no-fast-line", action="store_true", help="lo4 headline: skip the additional measurement of the fast schedule on the same sample")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of the other BASELINE configurations appended to the default line")
    ap.add_argument("--cpu-tflops", type=float, default=0.0, help="(internal) host TFLOP/s measured by the parent run: the cpu_baseline of an other_configs child")
    ap.add_argument("--lo4-vit", type=int, default=None, choices=[0, 1],
                    help="--precision lo4: 0 = the correction phase on the LLM layer linears only, 1 = on the SigLIP layer linears too; default = the "
                         "engine's 'auto' (the tower is corrected for samples whose LLM sequence is short, <= 1024 rows: C1; not for C2 / C3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-tp", action="store_true", help="N > 1: skip the additional one-sample-on-all-ranks (strong scaling) measurement")
    ap.add_argument("--tp-timeout", type=float, default=150.0, help="watchdog of the additional tensor-parallel measurement, seconds")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # LMI_BENCH_ONE_DEVICE=1: every rank on cuda:0 with a gloo group — a functional check of the multi-rank paths on a 1-GPU box
    one_device = os.environ.get("LMI_BENCH_ONE_DEVICE") == "1"
    dev = torch.device("cuda:0" if one_device else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    from leopard_amd import dist as D
    if world > 1:
        if not one_device:
            # the one-sample-on-all-ranks figure must come from RCCL through the C ABI (backend "rccl (lmi_comm)", rccl_ranks == N) or
            # not at all: without this the fallback to torch.distributed's group would be silent but for one field of the line
            os.environ.setdefault("LMI_COMM_STRICT", "1")
        D.init(backend="gloo" if one_device else "nccl", device=dev)     # RCCL over xGMI

    from leopard_amd.engine import KVCache, LeopardEngine
    from leopard_amd.ops import Ops
    from leopard_amd.weights import EngineWeights, SynthSource
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    for kv in args.opt:                                            # library options are process-wide: set them before any workload
        k, v = kv.split("=")
        Ops().set_option(k, int(v))
    if args.dtype == "fp8" and (args.workload == "idefics2-c4" or args.parallelism == "tp"):
        raise SystemExit("--dtype fp8 is built for the Leopard-LLaVA replica path (llava-c3 / llava-c5)")
    if args.workload == "idefics2-c4":
        return bench_idefics2(args, dev, dtype, rank, world, D)
    if args.workload == "llava-c5":
        return bench_c5(args, dev, dtype, rank, world, D)
    cfg = full_config()
    ops = Ops()
    t0 = time.perf_counter()
    tp = args.parallelism == "tp" and world > 1
    from leopard_amd.gpu_tiler import GpuTiler
    if tp:                                       # headline = ONE sample on all ranks
        r = measure_tp(args, cfg, ops, dev, dtype, rank, world, D, GpuTiler(ops, dev))
        out = {"metric": metric_name(args), "value": r["value"], "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"{config_label(args)}: {args.images}x({args.width}x{args.height}) images, one sample per step on all ranks",
                          "parallelism": r["parallelism"]},
               "backend": r["backend"], "rccl_ranks": r["rccl_ranks"], "comm_bytes_per_step": r["comm_bytes_per_step_per_rank"] * world,
               "tp": r}
        if rank == 0:
            print(json.dumps(out), flush=True)
        D.barrier()
        D.reset_comm()
        import torch.distributed as dist
        dist.destroy_process_group()
        return
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, dev, dtype), dtype)
    torch.cuda.synchronize()
    eng = LeopardEngine(cfg, W, ops=ops, device=dev)
    eng.fuse_norm_rope = not args.no_fuse
    eng.graph_encode = args.graph_encode
    if args.split_operands:
        args.precision = "split"
    if args.precision is None:
        args.precision = DEFAULT_PRECISION if args.dtype == "f16" else "fast"
        if args.precision == "lo4" and not eng.lo4_supported():
            args.precision = "fast"                            # a model shape the lo4 schedule does not cover (advisor r05): only an EXPLICIT request fails
    eng.precision = args.precision
    if args.lo4_vit is not None:
        eng.lo4_vit = bool(args.lo4_vit)
    load_s = time.perf_counter() - t0

    class Ctx:
        pass
    from leopard_amd.gpu_tiler import GpuTiler
    gpu_tiler = GpuTiler(ops, dev)
    ctxs = []
    host_tiler_s = gpu_tiler_s = 0.0
    for j in range(args.inflight):
        c = Ctx()
        u8, ids_np, plan, tiler_s, raw = make_sample(cfg, args.images, args.width, args.height, seed=rank * 16 + j)
        host_tiler_s = max(host_tiler_s, tiler_s)
        gpu_tiler.tile_sample(raw)                                 # warm (tap tables, allocator)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        c.raw_host = raw
        c.raw = [torch.from_numpy(np.array(r)).to(dev) for r in raw]   # source pixels resident in HBM before the timed region (np.array: a writable copy)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        c.tiles, _ = gpu_tiler.tile_sample(c.raw)                  # the tiler on the GPU (a3-a5), from resident source pixels
        torch.cuda.synchronize()
        gpu_tiler_s = max(gpu_tiler_s, time.perf_counter() - t1)
        if not torch.equal(c.tiles.cpu(), torch.from_numpy(u8)):   # same pixels as the host (PIL) tiler, bit for bit
            raise SystemExit("GPU tiler differs from the host tiler")
        c.ids = torch.from_numpy(ids_np).reshape(1, -1)            # token ids stay host-side, like a tokenizer's output
        c.n_tiles = u8.shape[0]
        c.S = c.ids.shape[1] + c.n_tiles * (cfg.tokens_per_tile - 1)
        c.cache = KVCache(cfg, c.S, dtype, dev)
        c.stream = torch.cuda.Stream(device=dev) if args.inflight > 1 else torch.cuda.current_stream(dev)
        ctxs.append(c)
    n_tiles, S = ctxs[0].n_tiles, ctxs[0].S
    if args.dtype == "fp8":
        enable_fp8(eng, cfg, args)

    def step(with_tiler=True):
        """One pass of the hot path over one sample: tiler (a1-a5, on the GPU, from the resident source pixels) -> SigLIP ->
        projector -> merge -> Llama prefill -> last-position logits ("tiler -> logits of last position")."""
        out = None
        for c in ctxs:
            with torch.cuda.stream(c.stream):
                c.cache.length = 0
                tiles = gpu_tiler.tile_sample(c.raw)[0] if with_tiler else c.tiles
                out = eng.prefill(c.ids, tiles, cache=c.cache)
        return out

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    def timed(n_warm, n_steps):
        for _ in range(n_warm):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            r = step()
        barrier()
        el = D.max_over_ranks(time.perf_counter() - t0, dev)
        assert r.seq_len == S and torch.isfinite(r.logits_last).all()
        return el, r
    elapsed, res = timed(args.warmup, args.steps)
    ms_per_step = elapsed / args.steps * 1e3
    # the same step from ready-made tiles (what round 1 timed): the tiler's share of the headline
    n_ex = max(2, min(args.steps, 5))
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_ex):
        step(with_tiler=False)
    barrier()
    ms_excl_tiler = D.max_over_ranks(time.perf_counter() - t0, dev) / n_ex * 1e3
    # and from source pixels that start in HOST memory (what a caller holding decoded images hands over): the same step plus the
    # PCIe copy of the images — reported beside the headline, never as `value`
    host_raw = [[np.array(r) for r in c.raw_host] for c in ctxs]
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_ex):
        for c, hr in zip(ctxs, host_raw):
            with torch.cuda.stream(c.stream):
                c.cache.length = 0
                eng.prefill(c.ids, gpu_tiler.tile_sample(hr)[0], cache=c.cache)
    barrier()
    ms_from_host = D.max_over_ranks(time.perf_counter() - t0, dev) / n_ex * 1e3
    images_per_s = world * args.inflight * args.images * args.steps / elapsed
    fl = algorithmic_flops(cfg, n_tiles, S)

    out = {
        "metric": metric_name(args),
        "value": round(images_per_s, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "precision_mode": PRECISION_NOTE[args.precision],
        **({"dtype_detail": fp8_detail(args), "accuracy_note": FP8_ACCURACY_NOTE,
            "prefill_mfma_frac_note": "algorithmic FLOPs / time against the 2.5 PF 16-bit peak (mixed-precision step)"}
           if args.dtype == "fp8" else {}),
        "config": {"workload": f"{config_label(args)}: {args.images}x({args.width}x{args.height}) images -> {n_tiles} ViT inputs (364x364), "
                               f"{n_tiles * cfg.tokens_per_tile} visual tokens, S={S}; SigLIP-SO400M/14 (27L) + Llama-3.1-8B (32L) "
                               "prefill to last-token logits, KV cache written; synthetic seeded weights",
                   "samples_per_rank_per_step": args.inflight, "samples_in_flight_per_gpu": args.inflight,
                   "parallelism": f"sample-sharded x{world} (the reference's scheme: one process per GPU over dataset shards, no data-path collective)"},
        "backend": ((("rccl (torch.distributed nccl)" if D.backend_name() == "nccl" else D.backend_name()) + ": barrier + timing reduce only")
                    if world > 1 else "none"),
        "rccl_ranks": (world if world > 1 and D.backend_name() == "nccl" else 0), "comm_bytes_per_step": 0,
        "visual_tokens_per_s": round(world * args.inflight * n_tiles * cfg.tokens_per_tile * args.steps / elapsed, 1),
        "algorithmic_tflop_per_step": round(args.inflight * fl["total"] / 1e12, 2),
        "prefill_mfma_frac": round(args.inflight * fl["total"] / 1e12 / (elapsed / args.steps) / MFMA_PEAK_TFLOPS, 4),
        "timed_region": "tiler (GPU, from source pixels resident in HBM) -> ViT -> projector -> merge -> LLM prefill -> last-token logits",
        "ms_per_step_excl_tiler": round(ms_excl_tiler, 3),
        "ms_per_step_from_host_pixels": round(ms_from_host, 3),
        "images_per_s_from_host_pixels": round(world * args.inflight * args.images / (ms_from_host * 1e-3), 3),
        "host_tiler_ms_per_sample": round(host_tiler_s * 1e3, 1), "gpu_tiler_ms_per_sample": round(gpu_tiler_s * 1e3, 2),
        "weight_load_s": round(load_s, 1),
    }

    if rank == 0:
        out["parity"] = fixture_parity(args, ctxs[0], res if args.inflight == 1 else None)
    if rank == 0 and not args.no_roofline:
        def once():
            with torch.cuda.stream(ctxs[0].stream):
                ctxs[0].cache.length = 0
                eng.prefill(ctxs[0].ids, ctxs[0].tiles, cache=ctxs[0].cache)
        passes = min(args.steps, 3)
        rl = roofline_from_timer(ops, once, passes, args.dtype == "fp8")
        # HBM-side bytes per launch come from separate rocprofv3 --pmc passes over this very command (tools/collect_traffic.sh ->
        # tools/hbm_traffic.py).  The committed summary is stamped with the hash of the kernel sources it was measured on and is
        # reported only when that hash matches the sources in this tree: a stale summary is refused (traffic = null).
        traffic, traffic_note = None, "no PMC summary for these kernel sources (run tools/collect_traffic.sh on the GPU box)"
        tpath = os.path.join(REPO, "profiles", "gemm_hbm_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("kernel_source_hash") == kernel_source_hash():
                traffic, traffic_note = round(tj["hbm_bytes_per_launch"]), "profiles/gemm_hbm_traffic.json: " + tj["method"]
            else:
                traffic_note = f"profiles/gemm_hbm_traffic.json is stale (measured on kernel sources {tj.get('kernel_source_hash')}); refused"
        if args.dtype == "fp8":
            traffic, traffic_note = None, "the committed PMC summary describes the f16 step; not collected for the fp8 line"
        rl["traffic"], rl["traffic_source"] = traffic, traffic_note
        if args.precision == "lo4":
            # matrix-pipe time of the launches at the peak of each phase's operand type: the 16-bit pass at 2.5 PF + the fp4 phase (the same
            # M x N x K again: 2 M N K4 FLOP) at 10 PF — what MFMA-busy counters see; `frac` above prices only the ALGORITHMIC FLOPs
            share = rl.get("correction_phase_flops_share", 0.0)
            rl["frac_incl_correction_phase"] = round(rl["frac"] * (1.0 + share * MFMA_PEAK_TFLOPS / MFMA_PEAK_FP4_TFLOPS), 4)
            rl["correction_phase_note"] = ("the corrected launches of the family (correction_phase_flops_share of its FLOPs) also run K4 / 256 k-tiles of "
                                           "v_mfma_scale_f32_32x32x64_f8f6f4 on fp4 images (4 x the 16-bit rate, dense peak 10 PF): + 25 % matrix-pipe time "
                                           "on those launches that `frac` books as overhead")
        out["roofline"] = rl
    if args.precision == "lo4":
        tail = eng.lo4_tail_rows(S)                                  # rows that carry the correction (engine.lo4_rows); the fp4 k-tiles run on their 256-row tiles
        corrected_rows = min(S, (((S - 1) // 256) - ((S - tail) // 256) + 1) * 256)
        out["lo4_rows"] = {"policy": eng.lo4_rows, "rows_with_residual_images": tail, "rows_in_corrected_tiles": corrected_rows, "of": S,
                           "note": "the correction covers the trailing rows of each sequence — the rows whose logits are read; the other rows' "
                                   "hand-over roundings reach them only through the softmax average (DESIGN.md 2.1, profiles/r06_lo4_policy_study_*.txt)"}
        vit_corrected = eng.lo4_vit_tiles([n_tiles], [S]) is not None
        lo_fl = fl["llm_linear"] * corrected_rows / S + ((fl["vit"] - n_tiles * vit_attention_flops(cfg)) if vit_corrected else 0)
        out["lo4_rows"]["siglip_tower_corrected"] = vit_corrected
        out["matrix_pipe_frac"] = round((args.inflight * (fl["total"] / MFMA_PEAK_TFLOPS + lo_fl / MFMA_PEAK_FP4_TFLOPS) / 1e12) / (elapsed / args.steps), 4)
        out["matrix_pipe_frac_note"] = ("algorithmic FLOPs at the 2.5 PF 16-bit peak + the correction phase's FLOPs (every corrected layer linear once more, "
                                        f"{lo_fl / 1e12:.1f} TFLOP) at the 10 PF fp4 peak, over the step time; prefill_mfma_frac counts the algorithmic FLOPs only")
    if args.precision == "lo4" and not args.no_fast_line and args.dtype != "fp8":
        # the SAME sample on the fast schedule (one rounding per operand hand-over, no correction phase): the throughput ceiling of the
        # 16-bit path and what it costs in parity — both modes in one line
        eng.precision = "fast"
        f_elapsed, f_res = timed(max(1, min(args.warmup, 2)), args.steps)
        fast = {"precision_mode": PRECISION_NOTE["fast"], "value": round(world * args.inflight * args.images * args.steps / f_elapsed, 3), "unit": "images/s",
                "ms_per_step": round(f_elapsed / args.steps * 1e3, 3), "steps": args.steps,
                "prefill_mfma_frac": round(args.inflight * fl["total"] / 1e12 / (f_elapsed / args.steps) / MFMA_PEAK_TFLOPS, 4),
                "lo4_over_fast_time": round(elapsed / f_elapsed, 4)}
        if rank == 0:
            fast["parity"] = fixture_parity(args, ctxs[0], f_res if args.inflight == 1 else None)
            if not args.no_roofline:
                frl = roofline_from_timer(ops, once, min(args.steps, 3), False)
                fast["roofline"] = {k: frl[k] for k in ("bound", "achieved", "peak", "unit", "frac", "dominant", "launches_per_step", "avg_launch_ms", "gemm_ms_per_step")}
        out["fast_schedule"] = fast
        eng.precision = args.precision
    cpu_tflops = args.cpu_tflops
    if rank == 0 and world == 1 and args.no_cpu_baseline and args.cpu_tflops > 0:
        out["cpu_baseline"] = extrapolated_cpu_baseline(args.cpu_tflops, args.images, fl["total"])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        tflops, t = cpu_baseline_sample(cfg)                   # doubles as the warm-up of the timed C1 runs
        cpu_tflops = tflops
        del eng, W, ctxs
        torch.cuda.empty_cache()
        c1_s, c1_tf, c1_S = cpu_baseline_c1(cfg, ops, dev)
        out["cpu_baseline"] = {
            "value": round(1.0 / c1_s, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"CPU oracle (fp32 PyTorch restatement of the reference path, oracle/leopard_oracle.py) END-TO-END prefill of "
                      f"config C1 (1 x 336x336 image, S = {c1_S}, 27 + 32 layers, {c1_tf:.2f} TFLOP): median of 3 runs after 1 warm-up "
                      f"= {c1_s:.2f} s = {c1_tf / c1_s:.3f} TFLOP/s on {torch.get_num_threads()} host threads",
            "seconds_c1": round(c1_s, 3), "tflops_c1": round(c1_tf / c1_s, 4),
            "c3_extrapolated": {"value": round(args.images / (fl["total"] / 1e12 / tflops), 5), "unit": "images/s",
                                "sample": f"2 SigLIP layers x 2 tiles + 1 Llama layer at S=1024 ({t:.2f} s, {tflops:.3f} TFLOP/s), "
                                          "scaled to the C3 sample (140.1 TFLOP) by algorithmic FLOPs"}}
    if world > 1 and not args.no_tp and args.dtype == "fp8":
        out["tp"] = {"skipped": "the tensor-parallel layer runs the 16-bit schedules (fast / lo4)"}
    elif world > 1 and not args.no_tp:
        # The same sample on ALL ranks (strong scaling), reported beside the replica headline.  It is the part of this program that no
        # 1-GPU box can exercise, so every rank runs it in a CHILD process (own process group on its own port, run_tp_child): an RCCL
        # abort, a hang or a crash there costs the "tp" object, never the headline line.
        del eng, W
        ctxs.clear()
        torch.cuda.empty_cache()
        try:
            D.barrier()
            out["tp"] = run_tp_child(args, rank, world, D, dev)
        except Exception as e:                                      # noqa: BLE001  (reported, not swallowed: the line carries it)
            out["tp"] = {"error": repr(e)[:500]}
    default_line = (args.workload == "llava-c3" and (args.images, args.width, args.height) == (6, 1344, 896) and args.dtype == "f16" and args.inflight == 1)
    if rank == 0 and world == 1 and default_line and not args.no_other_configs:
        if "eng" in dir():
            del eng, W
            ctxs.clear()
        torch.cuda.empty_cache()
        if cpu_tflops <= 0:
            cpu_tflops = cpu_baseline_sample(cfg)[0]
        out["other_configs"] = run_other_configs(cpu_tflops)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        try:
            D.barrier()
            D.reset_comm()
            dist.destroy_process_group()
        except Exception:                                           # noqa: BLE001  (teardown only; the result line is out)
            pass


if __name__ == "__main__":
    main()

"""End-to-end orchestration test on CPU: LeopardEngine driven over the emulated kernels (tools/hipemu) vs the
CPU oracle, on a micro configuration that still satisfies the kernel shape rules (ViT width 1152 = 16 x 72,
LLM head_dim 128).  Checks weight preparation (fusing / padding / gate-up interleave), the launch sequence,
the merge plan, KV-cache decode and greedy generation."""
import numpy as np
import pytest
import torch

from leopard_amd.config import LeopardConfig, RopeScaling, TextConfig, VisionConfig
from leopard_amd.engine import KVCache, LeopardEngine, plan_merge
from leopard_amd.synth import synth_state_dict_numpy
from leopard_amd.weights import EngineWeights, SynthSource, TensorSource
from oracle import leopard_oracle as O
from tests.emu_util import emu_ops


def micro_config():
    return LeopardConfig(
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=1, num_attention_heads=16,
                                   image_size=28, patch_size=14),
        text_config=TextConfig(hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                               num_key_value_heads=1, vocab_size=256, rope_scaling=RopeScaling()),
        image_token_index=250)


def cut_at_eos(out: torch.Tensor, n_prompt: int, eos) -> torch.Tensor:
    """What generate(..., eos_token_id=eos) returns, from the eos-free greedy output of the same call: greedy decoding is deterministic and
    the stop rule only truncates (EVAL:448-452: the eos token itself is emitted) — saves re-running the prefill in the emulator."""
    ids = out[0].tolist()
    for j in range(n_prompt, len(ids)):
        if ids[j] in eos:
            return out[:, :j + 1]
    return out


@pytest.fixture(scope="module")
def setup():
    ops = emu_ops()
    cfg = micro_config()
    Wn = synth_state_dict_numpy(cfg)
    return ops, cfg, Wn


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
def test_engine_prefill_matches_oracle(setup, dtype, tol):
    ops, cfg, Wn = setup
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    u8 = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (3, 28, 28, 3), dtype=np.uint8))
    ids = torch.tensor([[5, 250, 9, 250, 250, 17, 33]])
    res = eng.prefill(ids, u8, all_logits=True, keep_parts=True)
    from leopard_amd.tiler import siglip_normalize
    pix = torch.from_numpy(siglip_normalize(u8.numpy()))
    Wt = O.weights_from_numpy(Wn)
    logits, parts = O.prefill_logits(ids, pix, Wt, cfg, return_parts=True)
    assert (res.parts["vit"].float().view(3, 4, -1) - parts["vit"]).abs().max() <= tol * 4
    assert (res.parts["visual_tokens"].view(3, 1, -1) - parts["visual_tokens"]).abs().max() <= tol * 2
    assert (res.parts["inputs_embeds"] - parts["inputs_embeds"][0]).abs().max() <= tol * 2
    assert res.seq_len == logits.shape[1] == 7
    assert (res.logits_all - logits[0]).abs().max() <= tol * 2
    assert (res.logits_last - logits[0, -1]).abs().max() <= tol * 2


def test_synth_source_equals_tensor_source(setup):
    ops, cfg, Wn = setup
    a = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", torch.float16), torch.float16)
    b = EngineWeights.build(cfg, TensorSource(Wn, "cpu", torch.float16), torch.float16)
    assert torch.equal(a.vit_layers[0].qkv_w, b.vit_layers[0].qkv_w)
    assert torch.equal(a.llm_layers[1].gu_w, b.llm_layers[1].gu_w)
    assert torch.equal(a.patch_w_fused, b.patch_w_fused) and a.patch_w_fused.shape[1] == 704         # 14 pixel rows x 48, padded to 64s
    assert a.vit_ff == 128 and a.vit_layers[0].fc1_w.shape == (128, 1152)
    assert torch.equal(a.lm_head, b.lm_head) and torch.equal(a.pos_emb, b.pos_emb)


def test_generate_matches_oracle_greedy(setup):
    ops, cfg, Wn = setup
    dtype = torch.float16
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    u8 = torch.from_numpy(np.random.default_rng(4).integers(0, 256, (1, 28, 28, 3), dtype=np.uint8))
    ids = torch.tensor([[7, 250, 11, 12]])
    out = eng.generate(ids, u8, max_new_tokens=3, eos_token_id=())
    from leopard_amd.tiler import siglip_normalize
    ref = O.greedy_generate(ids, torch.from_numpy(siglip_normalize(u8.numpy())), O.weights_from_numpy(Wn), cfg, 3)
    assert out.shape == (1, 7) and torch.equal(out, ref)


def test_plan_merge_matches_oracle_and_raises():
    rng = np.random.default_rng(0)
    for _ in range(50):
        n = int(rng.integers(1, 40))
        ids = rng.integers(0, 20, n)
        tpt = int(rng.integers(1, 6))
        k = int((ids == 7).sum())
        assert np.array_equal(plan_merge(ids, 7, k * tpt, tpt), O.merge_plan(ids, 7, k * tpt, tpt))
    with pytest.raises(ValueError, match="number of image tokens"):
        plan_merge(np.array([1, 7, 2]), 7, 8, 4)



def test_prefill_batch_equals_per_sample_prefill(setup):
    """Packed multi-sample prefill (BASELINE config C5 shape) == per-sample prefill, on the emulator at the micro config."""
    ops, cfg, _ = setup
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", torch.float16), torch.float16)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    rng = np.random.default_rng(5)
    samples = [(torch.tensor([[5, 250, 9, 250, 17]]), torch.from_numpy(rng.integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))),
               (torch.tensor([[7, 8, 9]]), None),
               (torch.tensor([[250, 3]]), torch.from_numpy(rng.integers(0, 256, (1, 28, 28, 3), dtype=np.uint8)))]
    logits, seq_lens = eng.prefill_batch(samples)
    t = cfg.tokens_per_tile                          # visual tokens that replace each image token
    assert seq_lens == [5 + 2 * (t - 1), 3, 2 + (t - 1)]
    for i, (ids, tiles) in enumerate(samples):
        one = eng.prefill(ids, tiles)
        assert torch.equal(one.logits_last, logits[i])


def test_generate_batch_decodes_the_batch_together_and_equals_per_sample_generate(setup):
    """f4 batched decode: one packed prefill, then ALL samples advance together through the pooled KV cache and the skinny-M
    projections (lmi_gemm_skinny / lmi_rope_qk_rows / lmi_attn_decode_pool) — same tokens as per-sample generate(), with samples of
    different lengths, one of them stopping early at its eos."""
    ops, cfg, _ = setup
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", torch.float16), torch.float16)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    assert eng._batch_decode_supported()
    rng = np.random.default_rng(6)
    samples = [(torch.tensor([[5, 250, 9, 250, 17]]), torch.from_numpy(rng.integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))),
               (torch.tensor([[7, 8, 9]]), None),
               (torch.tensor([[250, 3]]), torch.from_numpy(rng.integers(0, 256, (1, 28, 28, 3), dtype=np.uint8)))]
    free = [eng.generate(ids, tiles, max_new_tokens=5, eos_token_id=()) for ids, tiles in samples]
    eos = (int(free[1][0, samples[1][0].shape[1] + 1]),)             # sample 1's SECOND new token: it stops there, the others go on
    singles = [cut_at_eos(o, smp[0].shape[1], eos) for o, smp in zip(free, samples)]
    assert torch.equal(singles[1], eng.generate(*samples[1], max_new_tokens=5, eos_token_id=eos))      # the rule cut_at_eos restates
    calls = []
    body = eng._batch_decode_body
    eng._batch_decode_body = lambda st: (calls.append(st.B), body(st))[1]
    batch = eng.generate_batch(samples, max_new_tokens=5, eos_token_id=eos)
    assert calls and set(calls) == {3} and len(calls) <= 4          # one step per token for the WHOLE batch, not per sample
    for one, got in zip(singles, batch):
        assert torch.equal(one, got)
    assert singles[1].shape[1] < samples[1][0].shape[1] + 5         # the early stop really happened
    st = eng._batch_states[3]
    again = eng.generate_batch(samples, max_new_tokens=5, eos_token_id=eos)      # state (pool, buffers) reused, same result
    assert eng._batch_states[3] is st and all(torch.equal(a, b) for a, b in zip(batch, again))
    # the RMSNorms folded into the projections (default) vs as launches of their own: the same tokens, and only 2 norm launches per step
    norms = []
    rms = ops.rmsnorm
    ops.rmsnorm = lambda *a, **k: (norms.append(1), rms(*a, **k))[1]
    try:
        calls.clear(); norms.clear()
        short = eng.generate_batch(samples, max_new_tokens=3, eos_token_id=())
        folded = len(norms) / len(calls)
        eng.skinny_fold_norm = False
        calls.clear(); norms.clear()
        plain = eng.generate_batch(samples, max_new_tokens=3, eos_token_id=())
        unfolded = len(norms) / len(calls)
    finally:
        ops.rmsnorm = rms
        eng.skinny_fold_norm = True
    assert all(torch.equal(a, b) for a, b in zip(short, plain))
    n_layers = len(W.llm_layers)
    assert unfolded - folded == 2 * n_layers - 1, (folded, unfolded)


def test_generate_stream_continuous_batching_keeps_slots_busy_and_equals_generate(setup):
    """f4 continuous batching (LeopardEngine.generate_stream): 7 samples of mixed lengths through 3 decode slots — a slot whose sequence
    ends (its own eos, or max_new_tokens) takes the next pending sample without re-creating the state; the stop rule runs on the device
    (live mask, token budget, eos ids), the host reads the tokens once per HIST steps; a sample whose FIRST token is an eos never takes a
    slot.  Every output equals the per-sample generate(), and the slot occupancy beats fixed groups of 3."""
    ops, cfg, _ = setup
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", torch.float16), torch.float16)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    eng.HIST = 2                                                     # short windows: admissions happen inside the test's few steps
    rng = np.random.default_rng(16)
    def img(n):
        return torch.from_numpy(rng.integers(0, 256, (n, 28, 28, 3), dtype=np.uint8))
    samples = [(torch.tensor([[5, 250, 9, 250, 17]]), img(2)), (torch.tensor([[7, 8, 9]]), None), (torch.tensor([[21, 3]]), None),
               (torch.tensor([[11, 12, 13, 14, 15, 16]]), None), (torch.tensor([[9, 250]]), img(1)), (torch.tensor([[33]]), None),
               (torch.tensor([[4, 5, 44, 6]]), None)]                 # the emulated ViT is the expensive part of a prefill here: two samples carry images
    T = 5
    free = [eng.generate(ids, tiles, max_new_tokens=T, eos_token_id=()) for ids, tiles in samples]
    # eos ids: sample 1's second new token (stops early) and sample 5's FIRST new token (finished by its prefill)
    eos = (int(free[1][0, samples[1][0].shape[1] + 1]), int(free[5][0, samples[5][0].shape[1]]))
    singles = [cut_at_eos(o, smp[0].shape[1], eos) for o, smp in zip(free, samples)]
    assert singles[5].shape[1] == samples[5][0].shape[1] + 1 and singles[1].shape[1] < samples[1][0].shape[1] + T
    states_before = dict(getattr(eng, "_batch_states", {}))
    stats = {}
    got = eng.generate_stream(samples, batch_size=3, max_new_tokens=T, eos_token_id=eos, stats=stats)
    for one, out in zip(singles, got):
        assert torch.equal(one, out), (one.tolist(), out.tolist())
    st = eng._batch_states[3]
    assert 3 not in states_before
    # occupancy: live slot-steps / slot-steps; fixed groups [0-2], [3-5], [6] would run max-length steps for every member
    assert stats["batch_size"] == 3 and stats["live_slot_steps"] == sum(o.shape[1] - s[0].shape[1] - 1 for o, s in zip(singles, samples))
    assert stats["live_slot_steps"] / stats["slot_steps"] > 0.55
    # state reused by a second stream (nothing re-created), same outputs; release drops the pools and the packed weight copy
    again = eng.generate_stream(list(reversed(samples)), batch_size=3, max_new_tokens=T, eos_token_id=eos)
    assert eng._batch_states[3] is st and all(torch.equal(a, b) for a, b in zip(reversed(singles), again))
    eng.release_batch_state()
    assert not eng._batch_states and eng._skinny_pack is None


def test_split_operand_mode_removes_the_hand_over_roundings(setup):
    """engine.split_operands: every A operand of every layer linear handed over as hi + lo (lmi_split_hi_lo, GEMMs at 2 K against [W | W],
    fp32 hand-overs from the norms / attention / GELU / SwiGLU).  The logits must sit where the oracle that treats exactly those sites as
    exact predicts — several times closer to fp32 than the production schedule — and the mode must not touch the production path."""
    ops, cfg, Wn = setup
    dtype = torch.float16
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    rng = np.random.default_rng(11)
    u8 = torch.from_numpy(rng.integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))
    ids = torch.tensor([[5, 250, 9, 250, 17, 33, 2]])
    from leopard_amd.tiler import siglip_normalize
    pix = torch.from_numpy(siglip_normalize(u8.numpy()))
    Wt = O.weights_from_numpy(Wn)
    ref = O.prefill_logits(ids, pix, Wt, cfg)[0]
    with O.emulate_rounding(dtype, exact_sites=("norm", "attn_out", "mlp_act")):
        emu = O.prefill_logits(ids, pix, Wt, cfg)[0]
    base = eng.prefill(ids, u8, all_logits=True).logits_all.clone()
    eng.split_operands = True
    cache = KVCache(cfg, 64, dtype, "cpu")
    res = eng.prefill(ids, u8, cache=cache, all_logits=True)
    got = res.logits_all
    scale = ref.abs().max().item()
    e_base = (base - ref).abs().max().item() / scale
    e_split = (got - ref).abs().max().item() / scale
    e_pred = (emu - ref).abs().max().item() / scale
    # at this depth (1 + 2 layers) the roundings the mode leaves in place (pixels, q / k / v, P, projector) still dominate; what is asserted
    # is that the mode lands on ITS predicted budget and below the production schedule (full depth: tests/test_gpu_parity.py)
    assert e_split < e_base, (e_split, e_base)
    assert 0.6 * e_pred <= e_split <= 1.5 * e_pred, (e_split, e_pred)
    assert cache.length == res.seq_len and bool(cache.k[1][:cache.length].abs().sum() > 0)          # K / V still appended by the q|k|v epilogue
    eng.split_operands = False
    assert torch.equal(eng.prefill(ids, u8, all_logits=True).logits_all, base)


def test_one_copy_of_the_llm_weights_serves_prefill_and_decode(setup):
    """LeopardEngine.pack_llm_weights (default): the layer linears live ONCE, in the operand order of the decode kernels; the prefill
    GEMM reads that order (ldw = LMI_LDW_PACKED(K)) bit-identically, batch-1 and batched decode stream the same tensors (no second
    copy is ever built), the natural-order q|k|v duplicate is gone and comes back from the rope-ordered rows on request."""
    from leopard_amd.weights import is_packed
    ops = setup[0]
    cfg = LeopardConfig(                                             # hidden 256: the fused schedule (folded norms, RoPE in the epilogue) applies
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=1, num_attention_heads=16, image_size=28, patch_size=14),
        text_config=TextConfig(hidden_size=256, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                               vocab_size=256, rope_scaling=RopeScaling()),
        image_token_index=250)
    mk = lambda: EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", torch.float16), torch.float16)
    W0, W1 = mk(), mk()
    plain = LeopardEngine(cfg, W0, ops=ops, device="cpu", pack_llm_weights=False)
    eng = LeopardEngine(cfg, W1, ops=ops, device="cpu")
    assert eng.llm_packed and not plain.llm_packed
    L0, L1 = W0.llm_layers[1], W1.llm_layers[1]
    assert L1.qkv_w is None and all(is_packed(getattr(L1, n)) for n in ("qkv_w_rope", "o_w", "gu_w", "down_w")) and not is_packed(W1.lm_head)
    assert torch.equal(eng._qkv_natural(L1), L0.qkv_w)
    layer_bytes = lambda W: sum(t.numel() * t.element_size() for L in W.llm_layers for t in (L.qkv_w, L.qkv_w_rope, L.o_w, L.gu_w, L.down_w) if t is not None)
    assert layer_bytes(W1) < layer_bytes(W0)
    rng = np.random.default_rng(26)
    u8 = torch.from_numpy(rng.integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))
    ids = torch.tensor([[5, 250, 9, 250, 17, 33]])
    a, b = plain.prefill(ids, u8, all_logits=True), eng.prefill(ids, u8, all_logits=True)
    assert torch.equal(a.logits_all, b.logits_all)                   # prefill: the same bits from either layout
    plain.fuse_norm_rope = eng.fuse_norm_rope = False                # un-fused schedule: q|k|v from the rope-ordered rows (RoPE on the fp32 sums)
    a2, b2 = plain.prefill(ids, u8), eng.prefill(ids, u8)
    assert (a2.logits_last - b2.logits_last).abs().max() <= 2e-3 * float(a2.logits_last.abs().max())
    plain.fuse_norm_rope = eng.fuse_norm_rope = True
    # decode: batch 1 runs on the batched-decode kernels with one row — tokens equal the GEMV step's and generate_batch's
    samples = [(ids, u8), (torch.tensor([[7, 8, 9]]), None)]
    singles = [eng.generate(s_ids, s_tiles, max_new_tokens=4, eos_token_id=()) for s_ids, s_tiles in samples]
    for (s_ids, s_tiles), one in zip(samples, singles):
        assert torch.equal(plain.generate(s_ids, s_tiles, max_new_tokens=4, eos_token_id=()), one)
    outs = eng.generate_batch(samples, max_new_tokens=4, eos_token_id=())
    assert all(torch.equal(o, one) for o, one in zip(outs, singles))
    assert getattr(eng, "_skinny_pack", None) is None                # nothing was copied for the batched step
    # precision modes read row-major views of the same tensors
    eng.split_operands = plain.split_operands = True
    assert torch.equal(plain.prefill(ids, u8).logits_last, eng.prefill(ids, u8).logits_last)
    eng.split_operands = plain.split_operands = False
    # and back
    eng.unpack_llm_weights()
    assert not eng.llm_packed and all(torch.equal(getattr(L1, n), getattr(L0, n)) for n in ("qkv_w", "qkv_w_rope", "o_w", "gu_w", "down_w"))
    assert torch.equal(eng.prefill(ids, u8, all_logits=True).logits_all, a.logits_all)


def test_a_packed_weight_that_lost_its_mark_is_refused(setup):
    """The layout of a weight is a mark on the tensor object; a copy does not carry it.  The engine remembers that it packed and refuses such a weight
    set instead of reading a packed matrix as row-major."""
    ops, cfg, _ = setup
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", torch.float16), torch.float16)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    assert eng.llm_packed
    for L in W.llm_layers:
        L.o_w = L.o_w.clone()                                        # the mark is gone, the bytes are still in the packed order
    with pytest.raises(RuntimeError, match="layout mark"):
        eng.prefill(torch.tensor([[7, 8, 9]]), None)


def test_vit_weights_in_the_packed_order_give_the_same_bits(setup):
    """LeopardEngine.pack_vit_weights (an A/B knob): the SigLIP linears read from the packed order by the same GEMM — identical features and logits,
    also through the split-operand mode; and back."""
    ops, cfg, _ = setup
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", torch.float16), torch.float16)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    u8 = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))
    ids = torch.tensor([[5, 250, 9, 250, 17]])
    a = eng.prefill(ids, u8, keep_parts=True)
    assert eng.pack_vit_weights() == 4 * len(W.vit_layers)
    b = eng.prefill(ids, u8, keep_parts=True)
    assert torch.equal(a.parts["vit"], b.parts["vit"]) and torch.equal(a.logits_last, b.logits_last)
    eng.split_operands = True
    s1 = eng.prefill(ids, u8).logits_last.clone()
    eng.split_operands = False
    assert eng.pack_vit_weights(False) == 4 * len(W.vit_layers)
    eng._split_w = None
    eng.split_operands = True
    assert torch.equal(eng.prefill(ids, u8).logits_last, s1)
    eng.split_operands = False
    assert torch.equal(eng.prefill(ids, u8).logits_last, a.logits_last)

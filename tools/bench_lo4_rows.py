"""Isolated-launch timing of the row selection of the correction phase (round 6): one Llama layer's four GEMMs at M = 7187 — the fast launch, lo4 on
every row, and lo4 with the last 256 rows selected: in the plain tile order, with the selected row tiles dispatched first (lmi_lo4.sel_ranges), and
with a ragged last row tile left in place (gemm.sel_ragged_last).  GPU only:  python tools/bench_lo4_rows.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from leopard_amd import _lib
from leopard_amd.ops import Ops, Lo4Act
from leopard_amd.weights import as_packed, interleave_gate_up, rope_permute_rows
dev = torch.device("cuda:0"); ops = Ops(); dt = torch.float16
M = 7187
g = torch.Generator(device=dev).manual_seed(1)
def sel_of(rows):
    row = torch.zeros(M, dtype=torch.uint8); row[list(rows)] = 1
    unit = torch.zeros((M + 63) // 64 * 64, dtype=torch.uint8); unit[:M] = row
    r = np.flatnonzero(np.diff(np.concatenate([[0], row.numpy().astype(np.int8), [0]])))
    return row.to(dev), unit.view(-1, 64).max(dim=1).values.contiguous().to(dev), np.ascontiguousarray(r.reshape(-1, 2).astype(np.int32))
def act_of(x, sel, ranges=True):
    a = Lo4Act.empty(x.shape[0], x.shape[1], dt, dev); ops.split_lo4(x, a)
    if sel is not None:
        a.row_sel, a.unit_sel = sel[0], sel[1]
        a.sel_ranges = sel[2] if ranges else None
        keep = sel[0].bool(); a.img[~keep] = 0; a.sc[~keep] = 0
    return a
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
H, KV, hd, D, F = 32, 8, 128, 4096, 14336
pos = torch.arange(M, device=dev).float(); inv = 1.0 / (5e5 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
cos, sin = (pos[:, None] * inv[None]).cos().contiguous(), (pos[:, None] * inv[None]).sin().contiguous()
ROWS = int(os.environ.get("ROWS", "256"))
sel = sel_of(range(M - ROWS, M))
shapes = {"q|k|v": ((H + 2 * KV) * hd, D), "o_proj": (D, D), "gate/up": (2 * F, D), "down": (D, F)}
print(f"{'':10s} {'fast':>9s} {'lo4 all':>9s} {'sel plain':>10s} {'sel first':>10s} {'sel first, ragged last':>22s}   (us per launch, M = {M}, last {ROWS} rows selected)")
for name, (N, K) in shapes.items():
    x = torch.randn(M, K, generator=g, device=dev)
    w = (torch.randn(N, K, generator=g, device=dev) * 0.02).to(dt)
    if name == "gate/up": w = interleave_gate_up(w[:N // 2].contiguous(), w[N // 2:].contiguous())
    if name == "q|k|v": w = torch.cat([rope_permute_rows(w[:(H + KV) * hd], hd), w[(H + KV) * hd:]], 0).contiguous()
    w4 = ops.quantize_w4(w); wp = as_packed(w)
    a_all, a_plain, a_first = act_of(x, None), act_of(x, sel, ranges=False), act_of(x, sel, ranges=True)
    gam = torch.ones(N, device=dev)
    def run(a, lo4=True):
        s = None if a is a_all else (sel if a.sel_ranges is not None else sel[:2])
        if name == "q|k|v":
            qkv = run.buf.setdefault("qkv", torch.empty(M, N, dtype=dt, device=dev))
            return (ops.rmsnorm_rope_lo4(a, wp, w4, qkv, None, 1e-5, cos, sin, None, None, 0, H, KV, hd) if lo4
                    else ops.rmsnorm_rope(a.hi, wp, qkv, None, 1e-5, cos, sin, None, None, 0, H, KV, hd))
        if name == "gate/up":
            o = run.buf.setdefault(("gu", id(a)), Lo4Act.empty(M, N // 2, dt, dev, sel=s))
            return (ops.gemm_lo4(a, wp, w4, o.hi, epilogue=_lib.EPI_SWIGLU, out4=o) if lo4 else ops.gemm_ex(a.hi, wp, o.hi, epilogue=_lib.EPI_SWIGLU))
        xs = run.buf.setdefault("xs", torch.zeros(M, N, device=dev)); sq = run.buf.setdefault("sq", torch.empty(M, N // 64, device=dev))
        h = run.buf.setdefault(("h", id(a)), Lo4Act.empty(M, N, dt, dev, sel=s))
        return (ops.gemm_lo4(a, wp, w4, xs, epilogue=_lib.EPI_RESIDUAL, norm_out=h.hi, norm_gamma=gam, rowsq_out=sq, out4=h) if lo4
                else ops.gemm_ex(a.hi, wp, xs, epilogue=_lib.EPI_RESIDUAL, norm_out=h.hi, norm_gamma=gam, rowsq_out=sq))
    run.buf = {}
    r = [t(lambda: run(a_all, False)), t(lambda: run(a_all)), t(lambda: run(a_plain)), t(lambda: run(a_first))]
    ops.set_option("gemm.sel_ragged_last", 1)
    r.append(t(lambda: run(a_first)))
    ops.set_option("gemm.sel_ragged_last", 0)
    r.append(t(lambda: run(a_all, False)))
    print(f"{name:10s} {r[0]:9.1f} {r[1]:9.1f} {r[2]:10.1f} {r[3]:10.1f} {r[4]:22.1f}   (fast again: {r[5]:.1f})")

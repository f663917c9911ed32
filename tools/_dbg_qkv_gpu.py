import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.test_gpu_lowbit import _selection, rnd, DEV
from leopard_amd.ops import Ops, Lo4Act
from leopard_amd.weights import as_packed, rope_permute_rows
ops = Ops(); dtype = torch.float16
S, D, H, KV, hd = 7187, 4096, 32, 8, 128
sel = _selection(S, range(S - 256, S)); keep = sel[0].bool()
x = rnd((S, D), torch.float32, 95, 2.0)
g = (torch.rand(D, generator=torch.Generator().manual_seed(96)) + 0.5).to(DEV)
full, part = Lo4Act.empty(S, D, dtype, DEV), Lo4Act.empty(S, D, dtype, DEV, sel=sel)
ops.norm_lo4(x, g, None, full, 1e-5); ops.norm_lo4(x, g, None, part, 1e-5)
w = rnd(((H + 2 * KV) * hd, D), dtype, 97, 0.05)
w_rope = torch.cat([rope_permute_rows(w[:(H + KV) * hd], hd), w[(H + KV) * hd:]], 0).contiguous()
w4 = ops.quantize_w4(w_rope); wp = as_packed(w_rope)
pos = torch.arange(S, device=DEV).float(); inv = 1.0 / (5e5 ** (torch.arange(0, hd, 2, device=DEV).float() / hd))
cos, sin = (pos[:, None] * inv[None]).cos().contiguous(), (pos[:, None] * inv[None]).sin().contiguous()
def run(mode):
    qkv = torch.empty(S, (H + 2 * KV) * hd, dtype=dtype, device=DEV)
    kc, vc = torch.zeros(S, KV * hd, dtype=dtype, device=DEV), torch.zeros(S, KV * hd, dtype=dtype, device=DEV)
    if mode == "fast": ops.rmsnorm_rope(full.hi, wp, qkv, None, 1e-5, cos, sin, kc, vc, 0, H, KV, hd)
    else:
        a = full if mode == "full" else part
        a.sel_ranges = sel[2] if mode == "sel" else None
        ops.rmsnorm_rope_lo4(a, wp, w4, qkv, None, 1e-5, cos, sin, kc, vc, 0, H, KV, hd)
    torch.cuda.synchronize()
    return (qkv, kc, vc)
o = {m: run(m) for m in ("fast", "full", "sel", "sel_noranges")}
for m in ("sel", "sel_noranges"):
    for i in range(3):
        a, b, c = o[m][i], o["full"][i], o["fast"][i]
        bad_k = (a[keep] != b[keep]).any(dim=1).nonzero().flatten()
        bad_u = (a[~keep] != c[~keep]).any(dim=1).nonzero().flatten()
        print(m, i, "bad keep rows", bad_k.numel(), bad_k[:8].tolist(), "bad other rows", bad_u.numel(), bad_u[:8].tolist(), bad_u[-4:].tolist() if bad_u.numel() else "")
        if bad_u.numel():
            r = int(bad_u[0]); cols = (a[~keep][r] != c[~keep][r]).nonzero().flatten()
            print("   first bad row cols", cols.numel(), cols[:6].tolist(), cols[-3:].tolist(), "nan", int(torch.isnan(a.float()).sum()))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for m in ("fast", "full", "sel", "sel_noranges"):
    print(m, round(t(lambda: run(m)), 1), "us (incl. allocs)")

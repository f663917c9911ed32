"""Isolated launch time of the prefill linears at the small BASELINE configurations (C1: S = 228 / 676 ViT rows; C4: S = 312; C2: S = 1242 / 4732 ViT rows)
under every tile geometry (gemm.config) and under the automatic choice.  GPU only:  python tools/bench_small_m_geometries.py  -> profiles/r06_small_m_gemm_sweep.txt"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from leopard_amd import _lib
from leopard_amd.ops import Ops
from leopard_amd.weights import as_packed
dev = torch.device("cuda:0"); ops = Ops(); dt = torch.float16
g = torch.Generator(device=dev).manual_seed(1)
def t(fn, n=30):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
CFGS = [(-1, "auto"), (0, "128x128"), (2, "256x128/3"), (4, "128x256/3"), (5, "256x256 st"), (8, "64x128/3"), (9, "64x128/6"), (10, "384x128")]
def shape(name, M, N, K, epi, act=_lib.ACT_NONE, packed=True):
    a = (torch.randn(M, K, generator=g, device=dev) * 0.5).to(dt)
    w0 = (torch.randn(N, K, generator=g, device=dev) * 0.02).to(dt)
    w = as_packed(w0) if packed else w0
    resid = epi == _lib.EPI_RESIDUAL
    out = torch.zeros(M, N // 2 if epi == _lib.EPI_SWIGLU else N, dtype=torch.float32 if resid else dt, device=dev)
    cells = []
    for cfg, label in CFGS:
        if cfg == 10 and M > 384: cells.append("      -"); continue
        ops.set_option("gemm.config", cfg)
        try:
            cells.append(f"{t(lambda: ops.gemm(a, w, out, epilogue=epi, act=act)):7.1f}")
        except RuntimeError:
            cells.append("    err")
    ops.set_option("gemm.config", -1)
    mb = N * K * 2 / 1e6
    print(f"{name:<34} M={M:5d} N={N:6d} K={K:6d} W={mb:6.1f} MB | " + " ".join(cells), flush=True)
print(" " * 71 + "| " + " ".join(f"{l:>7.7s}" for _, l in CFGS))
for S in (228, 312, 1242):
    shape(f"Llama q|k|v (S={S})", S, 6144, 4096, _lib.EPI_STORE)
    shape(f"Llama o_proj (S={S})", S, 4096, 4096, _lib.EPI_RESIDUAL)
    shape(f"Llama gate/up (S={S})", S, 28672, 4096, _lib.EPI_SWIGLU)
    shape(f"Llama down (S={S})", S, 4096, 14336, _lib.EPI_RESIDUAL)
for M in (676, 4732):
    shape(f"SigLIP q|k|v (M={M})", M, 3456, 1152, _lib.EPI_STORE, packed=False)
    shape(f"SigLIP out_proj (M={M})", M, 1152, 1152, _lib.EPI_RESIDUAL, packed=False)
    shape(f"SigLIP fc1 (M={M})", M, 4352, 1152, _lib.EPI_STORE, _lib.ACT_GELU_TANH, packed=False)
    shape(f"SigLIP fc2 (M={M})", M, 1152, 4352, _lib.EPI_RESIDUAL, packed=False)

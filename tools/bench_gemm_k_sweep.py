"""Fixed cost per output tile of the prefill GEMM: time(K) at fixed M, N is a + b K; a / tiles-per-CU is what a tile pays outside its k-loop
(launch ramp, ring fill, epilogue, drain).  GPU only:  python tools/bench_gemm_k_sweep.py   (profiles/r05_gemm_k_sweep.txt)"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from leopard_amd import _lib
from leopard_amd.ops import Ops
from leopard_amd.weights import as_packed
dev = torch.device("cuda:0"); ops = Ops(); dt = torch.float16
for kv in filter(None, os.environ.get("LMI_OPTS", "").split(",")):      # e.g. LMI_OPTS=gemm.persist=0
    k, v = kv.split("="); ops.set_option(k, int(v)); print("# option", k, v)
g = torch.Generator(device=dev).manual_seed(1)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def sweep(name, M, N, epilogue, act, resid):
    rows = []
    for K in (256, 640, 1152, 2304, 4608):
        a = (torch.randn(M, K, generator=g, device=dev) * 0.5).to(dt)
        w = as_packed((torch.randn(N, K, generator=g, device=dev) * 0.02).to(dt))
        bias = torch.zeros(N, dtype=torch.float32, device=dev)
        out = torch.zeros(M, N, dtype=torch.float32 if resid else dt, device=dev)
        us = t(lambda: ops.gemm(a, w, out, bias=bias, epilogue=epilogue, act=act))
        rows.append((K, us))
    (k1, t1), (k2, t2) = rows[2], rows[4]
    b = (t2 - t1) / (k2 - k1); a0 = t1 - b * k1
    print(f"{name:<44} M = {M}, N = {N}: " + "  ".join(f"K={k}: {u:7.1f} us" for k, u in rows))
    print(f"{'':<44} fit on K = 1152 / 4608: fixed {a0:6.1f} us + {b * 64:5.2f} us per 64-deep k-tile -> fixed share at K = 1152: {a0 / t1:.0%}; "
          f"k-loop rate {2 * M * N / b * 1e-6:.0f} TFLOP/s")
M = 42 * 729
sweep("SigLIP q|k|v (store)", M, 3456, _lib.EPI_STORE, _lib.ACT_NONE, False)
sweep("SigLIP fc1 (store + GELU)", M, 4352, _lib.EPI_STORE, _lib.ACT_GELU_TANH, False)
sweep("SigLIP out_proj / fc2 shape (residual)", M, 1152, _lib.EPI_RESIDUAL, _lib.ACT_NONE, True)
sweep("Llama o_proj shape (residual)", 7187, 4096, _lib.EPI_RESIDUAL, _lib.ACT_NONE, True)
sweep("Llama q|k|v shape (store)", 7187, 6144, _lib.EPI_STORE, _lib.ACT_NONE, False)

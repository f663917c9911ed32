"""Is the GEMM epilogue bound per CU or by the shared memory system?  One round of 256 / 128 / 64 / 32 / 8 tiles at K = 256, with and without
the epilogue (LEOPARD_AMD_LIB = a -DLMI_EXP_NOEPI build for the second arm)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from leopard_amd import _lib
from leopard_amd.ops import Ops
dev = torch.device("cuda:0"); ops = Ops(); dt = torch.float16
ops.set_option("gemm.config", 5)
g = torch.Generator(device=dev).manual_seed(1)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for epi, name in ((_lib.EPI_STORE, "store"), (_lib.EPI_RESIDUAL, "resid")):
    for K in (256, 1152):
        row = []
        for tiles_m in (32, 16, 8, 4, 1):
            M, N = tiles_m * 256, 2048
            a = (torch.randn(M, K, generator=g, device=dev) * 0.5).to(dt)
            w = (torch.randn(N, K, generator=g, device=dev) * 0.02).to(dt)
            out = torch.zeros(M, N, dtype=torch.float32 if epi == _lib.EPI_RESIDUAL else dt, device=dev)
            row.append((tiles_m * 8, t(lambda: ops.gemm(a, w, out, epilogue=epi))))
        print(f"{name} K={K}: " + "  ".join(f"{n} tiles {u:6.1f} us" for n, u in row), flush=True)

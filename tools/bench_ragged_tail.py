"""What the ragged last row tile of the C3 sequence costs: the four Llama linears at M = 7168 (28 whole 256-row tiles), 7187 (the C3 length: + a 19-row
tile) and 7424 (29 whole tiles), isolated launches of the production kernels.  GPU only:  python tools/bench_ragged_tail.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from leopard_amd import _lib
from leopard_amd.ops import Ops
from leopard_amd.weights import as_packed
dev = torch.device("cuda:0"); ops = Ops(); dt = torch.float16
g = torch.Generator(device=dev).manual_seed(1)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print(f"{'':10s} " + " ".join(f"{'M=' + str(m):>10s}" for m in (7168, 7187, 7424)) + "   (us per launch; 7187 - 7168 = the 19-row tile, 7424 - 7168 = a whole row of tiles)")
for name, N, K, epi in (("q|k|v", 6144, 4096, _lib.EPI_STORE), ("o_proj", 4096, 4096, _lib.EPI_RESIDUAL), ("gate/up", 28672, 4096, _lib.EPI_SWIGLU), ("down", 4096, 14336, _lib.EPI_RESIDUAL)):
    w = as_packed((torch.randn(N, K, generator=g, device=dev) * 0.02).to(dt))
    row = []
    for M in (7168, 7187, 7424):
        a = (torch.randn(M, K, generator=g, device=dev) * 0.5).to(dt)
        out = torch.zeros(M, N // 2 if epi == _lib.EPI_SWIGLU else N, dtype=torch.float32 if epi == _lib.EPI_RESIDUAL else dt, device=dev)
        row.append(t(lambda: ops.gemm(a, w, out, epilogue=epi)))
    print(f"{name:10s} " + " ".join(f"{u:10.1f}" for u in row))

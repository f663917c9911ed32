"""Isolated-launch timing of one Llama layer's gate/up and down GEMMs: fast schedule, lo4 without and with the residual image written
(profiles/r05_lo4_epilogue_ab.txt).  GPU only:  python tools/bench_lo4_launch.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from leopard_amd import _lib
from leopard_amd.ops import Ops, Lo4Act
from leopard_amd.weights import as_packed, interleave_gate_up
dev = torch.device("cuda:0"); ops = Ops(); dt = torch.float16
M, F, K = 7187, 14336, 4096
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(M, K, generator=g, device=dev)
w = interleave_gate_up((torch.randn(F, K, generator=g, device=dev) * 0.02).to(dt), (torch.randn(F, K, generator=g, device=dev) * 0.02).to(dt))
act = Lo4Act.empty(M, K, dt, dev); ops.split_lo4(x, act)
w4 = ops.quantize_w4(w); wp = as_packed(w)
sq = (torch.rand(M, K // 64, generator=g, device=dev) + 0.5) * 64
out4 = Lo4Act.empty(M, F, dt, dev); out4.img.zero_(); out4.sc.zero_()
plain = torch.empty(M, F, dtype=dt, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
res = {}
res["fast (16-bit only)"] = t(lambda: ops.gemm_ex(act.hi, wp, plain, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=K, norm_eps=1e-5))
res["lo4, no image out"] = t(lambda: ops.gemm_lo4(act, wp, w4, plain, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=K, norm_eps=1e-5))
res["lo4, image out"] = t(lambda: ops.gemm_lo4(act, wp, w4, out4.hi, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=K, norm_eps=1e-5, out4=out4))
wd = (torch.randn(K, F, generator=g, device=dev) * 0.02).to(dt)
wd4 = ops.quantize_w4(wd); wdp = as_packed(wd)
xres = torch.zeros(M, K, device=dev); gam = torch.ones(K, dtype=dt, device=dev)
hN = Lo4Act.empty(M, K, dt, dev); sqo = torch.empty(M, K // 64, device=dev)
res["down fast"] = t(lambda: ops.gemm_ex(out4.hi, wdp, xres, epilogue=_lib.EPI_RESIDUAL, norm_out=hN.hi, norm_gamma=gam, rowsq_out=sqo))
res["down lo4, no image out"] = t(lambda: ops.gemm_lo4(out4, wdp, wd4, xres, epilogue=_lib.EPI_RESIDUAL, norm_out=hN.hi, norm_gamma=gam, rowsq_out=sqo))
res["down lo4, image out"] = t(lambda: ops.gemm_lo4(out4, wdp, wd4, xres, epilogue=_lib.EPI_RESIDUAL, norm_out=hN.hi, norm_gamma=gam, rowsq_out=sqo, out4=hN))
print(os.environ.get("LEOPARD_AMD_LIB", "prod").split("/")[-1], {k: round(v, 1) for k, v in res.items()})

#!/usr/bin/env python3
"""Per-kernel HBM-side traffic of the C3 step (WHICH launches over-fetch, and by how much): mean FETCH_SIZE (x 2, KiB ->
bytes: tools/hbm_traffic.py) and WRITE_SIZE per launch of every GEMM / attention kernel, and — when a third counter file is given — the L2 hit
rate (TCC_HIT_sum / TCC_REQ_sum) per launch.  usage: traffic_by_kernel.py <fetch.csv> <write.csv> [<tcc.csv>]
Algorithmic bytes of the C3 shapes (M = 7187 Llama rows / 28392 SigLIP rows, 16-bit operands) are printed beside the GEMMs that have one shape per kernel name."""
import csv
import sys
from collections import defaultdict


def per_kernel(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


S, V = 7187, 28392
# kernel-name fragments (template arguments: EPI, ACT, AMODE, geometry) -> (label, algorithmic read bytes A + W, algorithmic written / RMW bytes)
SHAPES = [
    ("Li3ELi0ELi0ENS_7GemmCfgILi256ELi256", "Llama gate/up + SwiGLU", (S * 4096 + 28672 * 4096) * 2, S * 14336 * 2),
    ("Li4ELi0ELi0ENS_7GemmCfgILi256ELi256", "Llama q|k|v + RoPE", (S * 4096 + 6144 * 4096) * 2, S * 6144 * 2 + S * 2048 * 2),
    ("Li1ELi0ELi0ENS_7GemmCfgILi256ELi256", "Llama o_proj / down_proj (avg)", ((S * 4096 + 4096 * 4096) + (S * 14336 + 4096 * 14336)) * 2 // 2, S * 4096 * (4 + 4 + 2)),
    ("Li0ELi1ELi0ENS_7GemmCfgILi256ELi256", "SigLIP fc1 + GELU", (V * 1152 + 4352 * 1152) * 2, V * 4352 * 2),
    ("Li0ELi0ELi0ENS_7GemmCfgILi256ELi256", "SigLIP q|k|v", (V * 1152 + 3456 * 1152) * 2, V * 3456 * 2),
    ("Li1ELi0ELi0ENS_7GemmCfgILi256ELi128", "SigLIP fc2 (fp32 +=)", (V * 4352 + 1152 * 4352) * 2, V * 1152 * 8),
    ("Li1ELi0ELi0ENS_7GemmCfgILi128ELi128", "SigLIP out_proj (fp32 +=)", (V * 1152 + 1152 * 1152) * 2, V * 1152 * 8),
    ("attn_fwd_dma_kernelIDF16_Li128", "Llama attention (causal, S = 7187)", S * 6144 * 2, S * 4096 * 2),
    ("attn_fwd_dma_kernelIDF16_Li72", "SigLIP attention (42 x 676)", V * 3456 * 2, V * 1152 * 2),
    ("norm_kernel", "LayerNorm / RMSNorm", V * 1152 * 4, V * 1152 * 2),
]
fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
tcc = per_kernel(sys.argv[3]) if len(sys.argv) > 3 else {}
print(f"{'kernel':<36} {'launches':>8} {'fetch MB':>9} {'write MB':>9} | {'alg. read':>9} {'alg. out':>9} | {'fetch / alg.':>12} {'write / alg.':>12}" + ("   L2 hit" if tcc else ""))
for frag, label, rd, wr in SHAPES:
    fk = [k for k in fetch if frag in k]
    if not fk:
        continue
    fv = [v for k in fk for v in fetch[k]["FETCH_SIZE"]]
    wv = [v for k in write if frag in k for v in write[k]["WRITE_SIZE"]]
    f = sum(fv) / len(fv) * 2048
    w = sum(wv) / max(len(wv), 1) * 1024
    hit = ""
    if tcc:
        h = sum(v for k in tcc if frag in k for v in tcc[k].get("TCC_HIT_sum", []))
        q = sum(v for k in tcc if frag in k for v in tcc[k].get("TCC_REQ_sum", []))
        hit = f"   {h / q:6.1%}" if q else ""
    print(f"{label:<36} {len(fv):8d} {f / 1e6:9.1f} {w / 1e6:9.1f} | {rd / 1e6:9.1f} {wr / 1e6:9.1f} | {f / rd:12.2f} {w / wr:12.2f}{hit}")

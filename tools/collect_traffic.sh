#!/bin/bash
# HBM-side traffic of the GEMM family per launch (roofline.traffic of bench.py), run on the GPU box from the repo root:
#   bash tools/collect_traffic.sh      -> profiles/gemm_hbm_traffic.json (stamped with the kernel-source hash)
# Two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), counters only — never
# combined with tracing options (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -e
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/traffic
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-fast-line --no-other-configs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
python tools/hbm_traffic.py "$F" "$W" "$OUT/gemm_hbm_traffic.json"
cp "$OUT/gemm_hbm_traffic.json" profiles/gemm_hbm_traffic.json 2>/dev/null || true
# the raw CSVs are large: keep only the summary in gpurun_out
rm -rf "$OUT/fetch" "$OUT/write"

#!/bin/bash
# Per-kernel HBM-side traffic and L2 hit rate of the C3 step (run on the GPU box from the repo root) -> gpurun_out/traffic_by_kernel.txt
# Three separate rocprofv3 --pmc passes, counters only (never combined with tracing options).
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/tbk
rm -rf "$OUT"; mkdir -p "$OUT"
PREC=${1:-lo4}
CMD="python $R/bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-fast-line --no-other-configs --precision $PREC"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_REQ_sum --output-format csv -d "$OUT/tcc" -- $CMD > "$OUT/tcc.log" 2>&1
cd $R
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
T=$(find "$OUT/tcc" -name '*counter_collection.csv' | head -1)
python tools/traffic_by_kernel.py "$F" "$W" $T > $R/gpurun_out/traffic_by_kernel_$PREC.txt 2>&1
tail -3 "$OUT/tcc.log"
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/tcc"
cat $R/gpurun_out/traffic_by_kernel_$PREC.txt

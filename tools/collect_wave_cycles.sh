#!/bin/bash
# Wave-cycle accounting per kernel of the C3 step (run on the GPU box from the repo root) -> gpurun_out/wave_cycles_<precision>.txt
# One rocprofv3 --pmc pass (8 SQ slots), counters only.  WAIT_ANY (parked at s_waitcnt / s_barrier) + WAIT_INST_ANY (issue stall) +
# ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots).
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
PREC=${1:-lo4}
OUT=$R/gpurun_out/wc_$PREC
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT" -- \
  python $R/bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-fast-line --no-other-configs --precision $PREC > "$OUT/run.log" 2>&1
cd $R
F=$(find "$OUT" -name '*counter_collection.csv' | head -1)
python - "$F" > $R/gpurun_out/wave_cycles_$PREC.txt <<'P'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]
print(f"{'launches':>8} {'wave cycles':>13} {'parked':>7} {'issue stall':>11} {'active':>7} {'VALU':>6} {'LDS':>6} {'bank confl / LDS':>16}  kernel")
for k, c in rows:
    w = c.get("SQ_WAVE_CYCLES", 0) or 1.0
    lds = c.get("SQ_ACTIVE_INST_LDS", 0) or 1.0
    print(f"{n[k]:8d} {w:13.3e} {c.get('SQ_WAIT_ANY', 0) / w:7.1%} {c.get('SQ_WAIT_INST_ANY', 0) / w:11.1%} {c.get('SQ_ACTIVE_INST_ANY', 0) / w:7.1%} "
          f"{c.get('SQ_ACTIVE_INST_VALU', 0) / w:6.1%} {c.get('SQ_ACTIVE_INST_LDS', 0) / w:6.1%} {c.get('SQ_LDS_BANK_CONFLICT', 0) / lds:16.1%}  {k[:110]}")
P
tail -2 "$OUT/run.log" | cut -c1-200
rm -rf "$OUT"
cat $R/gpurun_out/wave_cycles_$PREC.txt

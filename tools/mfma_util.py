#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from a rocprofv3 counter pass
(`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv`): per kernel name, launches, mean
GRBM_GUI_ACTIVE, and utilisation = MFMA-busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) — the method of profiles/README.md (round 1).
usage: mfma_util.py <counter_collection.csv>"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    busy, act = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(d["GRBM_GUI_ACTIVE"])
    if busy == 0:
        continue
    rows.append((act, len(d["GRBM_GUI_ACTIVE"]), busy / (1024.0 * act / 8.0), k))
print(f"{'launches':>8s} {'GUI_ACTIVE total':>18s} {'MFMA util':>9s}  kernel")
for act, n, u, k in sorted(rows, reverse=True):
    print(f"{n:8d} {act:18.0f} {100 * u:8.1f}%  {k[:150]}")

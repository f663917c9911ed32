#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection.csv: mean counter value per kernel name.  usage: pmc_summary.py <csv> [filter]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if flt and flt not in k:
        continue
    acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")

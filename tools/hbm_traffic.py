#!/usr/bin/env python3
"""HBM-side bytes per launch of the GEMM family from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950).  usage: hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
The summary is stamped with the hash of the kernel sources it was measured on (bench.kernel_source_hash); bench.py reports it
as roofline.traffic only while that hash matches the tree.

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; FETCH_SIZE tallies the 128-byte requests of wide
(16 B/lane) streaming reads at 64 B, so it is doubled; WRITE_SIZE matched a known output size exactly (profiles/README.md) and
is taken as is.  The counters sit on the L2's memory side: Infinity-Cache hits are included."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
is_gemm = lambda k: "gemm_kernel" in k or "gemm_stagger_kernel" in k
launches = sum(len(v) for k, v in fetch.items() if is_gemm(k))
fetch_b = sum(sum(v) for k, v in fetch.items() if is_gemm(k)) * 1024 * 2
write_b = sum(sum(v) for k, v in write.items() if is_gemm(k)) * 1024
out = {"kernel_family": "lmi::gemm_kernel + lmi::gemm_stagger_kernel (all epilogues)", "launches_counted": launches,
       "kernel_source_hash": kernel_source_hash(),
       "fetch_bytes_per_launch": fetch_b / launches, "write_bytes_per_launch": write_b / launches,
       "hbm_bytes_per_launch": (fetch_b + write_b) / launches,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 1 --warmup 1 "
                 "--no-roofline --no-cpu-baseline --no-fast-line --no-other-configs` (the default schedule: lo4); KiB -> bytes, FETCH_SIZE x2 (gfx950 wide-read under-count), Infinity-Cache hits included"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))

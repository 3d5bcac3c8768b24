#!/usr/bin/env python3
"""Folds rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json (HBM bytes per launch).

gfx950 corrections (MI355X_MICROARCH.md §HBM): counters are in KiB; FETCH_SIZE under-reports wide coalesced
streaming reads by exactly 2x, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
Usage: python tools/pmc_traffic.py <dir-with-both-passes> > profiles/pmc_traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = row["Kernel_Name"]
        if "igemm_kernel" in name and "Lb1E" not in name and ", true," not in name:
            key = "linear"
        elif "igemm_kernel" in name:
            key = "roofline"            # gathered implicit GEMM = the dominant class of bench.py
        elif "temporal_attention_kernel" in name:
            key = "roofline_temporal"
        elif "attention_kernel" in name:
            key = "attention"
        else:
            continue
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
detail = {}
for key, cs in acc.items():
    fetch = cs.get("FETCH_SIZE", [])
    write = cs.get("WRITE_SIZE", [])
    if not fetch or not write:
        continue
    fb = 2.0 * 1024.0 * sum(fetch) / len(fetch)
    wb = 1024.0 * sum(write) / len(write)
    out[key] = fb + wb
    detail[key] = {"launches_fetch_pass": len(fetch), "launches_write_pass": len(write),
                   "fetch_bytes_per_launch_x2_corrected": fb, "write_bytes_per_launch": wb}
out["_detail"] = detail
out["_note"] = "HBM-side bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction), averaged over every launch of the class"
print(json.dumps(out, indent=1))

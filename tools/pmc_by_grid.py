#!/usr/bin/env python3
"""Per (kernel, grid) HBM-side traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gfx950 corrections of
MI355X_MICROARCH.md: KiB units, FETCH_SIZE x 2).  Usage: python tools/pmc_by_grid.py <dir> [substring]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = row["Kernel_Name"]
        if want not in name:
            continue
        short = re.sub(r"\(.*$", "", name).replace("void ", "").replace("lavie::", "")[:70]
        key = (short, row.get("Grid_Size", ""), row.get("Workgroup_Size", ""))
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
rows = []
for key, cs in acc.items():
    fe = cs.get("FETCH_SIZE", [])
    wr = cs.get("WRITE_SIZE", [])
    rd = 2.0 * 1024 * sum(fe) / len(fe) if fe else float("nan")
    wb = 1024.0 * sum(wr) / len(wr) if wr else float("nan")
    rows.append((rd + wb, key, len(fe) or len(wr), rd, wb))
print(f"{'kernel':70s} {'grid':>10s} {'n':>5s} {'read MB':>9s} {'write MB':>9s}")
for tot, key, n, rd, wb in sorted(rows, key=lambda r: -(r[0] * r[2]) if r[0] == r[0] else 0):
    print(f"{key[0]:70s} {key[1]:>10s} {n:5d} {rd / 1e6:9.1f} {wb / 1e6:9.1f}")

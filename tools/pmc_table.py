#!/usr/bin/env python3
"""Prints per-kernel averages of every counter found in rocprofv3 --pmc CSV output under a directory."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "lavie"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = row["Kernel_Name"]
        if want not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("lavie::", "")[:60]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} n={len(v):3d} avg={sum(v) / len(v):.4g}")

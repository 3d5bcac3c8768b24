#!/usr/bin/env python3
"""Folds a rocprofv3 --kernel-trace CSV into a per-(kernel, grid) table: launches, total/avg/min/max duration.
Usage: python tools/summarize_trace.py <dir-with-*_kernel_trace.csv> > profiles/rNN_kernel_summary.md"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("lavie::", "").replace("void ", "")
    return name[:110]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + root)
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    total = 0.0
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3      # us
                grid = "x".join(str(row.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
                wg = row.get("Workgroup_Size_X", "")
                key = (short(row["Kernel_Name"]), grid, wg)
                a = agg[key]
                a[0] += 1
                a[1] += dur
                a[2] = min(a[2], dur)
                a[3] = max(a[3], dur)
                total += dur
    print(f"# kernel trace summary ({len(files)} file(s)); total device time {total / 1e3:.2f} ms\n")
    # per-kernel totals over all grid sizes (what bench.py's roofline.avg_launch_us is checked against)
    per = defaultdict(lambda: [0, 0.0])
    for key, a in agg.items():
        k = re.sub(r"<.*$", "", key[0]) if not key[0].startswith("igemm") else key[0]
        per[k][0] += a[0]
        per[k][1] += a[1]
    print("| kernel (all grid sizes) | launches | total ms | avg us |")
    print("|---|---|---|---|")
    for k, a in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"| `{k}` | {a[0]} | {a[1] / 1e3:.3f} | {a[1] / a[0]:.1f} |")
    print()
    print("| kernel | grid (threads) | wg | launches | total ms | % | avg us | min us | max us |")
    print("|---|---|---|---|---|---|---|---|---|")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{key[0]}` | {key[1]} | {key[2]} | {a[0]} | {a[1] / 1e3:.3f} | {100 * a[1] / total:.1f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} |")


if __name__ == "__main__":
    main()

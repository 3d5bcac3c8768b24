#!/usr/bin/env python3
"""Folds rocprofv3 --pmc passes over bench.py into profiles/pmc_traffic.json.

* HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (separate passes).  gfx950 corrections (MI355X_MICROARCH.md §HBM):
  the counters are in KiB; FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x, so it is doubled;
  WRITE_SIZE is exact for 16-B-per-lane streaming stores.
* MFMA busy fraction from SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE when that pass is present:
  busy / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs), i.e. the share of SIMD cycles AT THE CLOCK THE CHIP HELD in which the
  matrix pipe was executing (GRBM_GUI_ACTIVE is summed over the 8 XCDs).
Keys: "roofline" = igemm_patch_kernel<..., MODE != 3> (bench.py's dominant kernel: the 3x3 instances), "upsample_parity" = its MODE 3 launches, "conv_class" = every gathered implicit GEMM,
"linear" = plain implicit GEMMs, "attention", "roofline_temporal", and (round 3) "roofline_fused_temporal" / "roofline_fused_feed_forward" / "roofline_fused_cross_attention"
= the row-resident fused sub-block kernels of rowfuse.hip.
Usage: python tools/pmc_traffic.py <dir-with-the-passes> > profiles/pmc_traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def classes(name):
    out = []
    if "igemm_patch_kernel" in name:
        # bench.py's `roofline` object = the 3x3 instances only (template MODE 0 / 1 / 2: last argument); the parity-form upsample
        # launches (MODE 3) have their own key.  Demangled "igemm_patch_kernel<0, 6, 5, 3>" or mangled "...ILi0ELi6ELi5ELi3EE..."
        parity = name.replace(" ", "").split(">")[0].endswith(",3") or "Li5ELi3EE" in name or "Li4ELi3EE" in name
        out += ["upsample_parity" if parity else "roofline", "conv_class"]
    elif "igemm_pp_kernel<true" in name or "igemm_pp_kernelILb1" in name:
        out.append("conv_class")
    elif "igemm_ppx_kernel" in name or "igemm_pp_kernel" in name:
        out.append("linear")
    elif "igemm_kernel" in name:
        gather = ", true," in name or "Lb1E" in name
        out.append("conv_class" if gather else "linear")
    elif "temporal_block_kernel" in name:
        out.append("roofline_fused_temporal")
    elif "geglu_mlp_kernel" in name:
        out.append("roofline_fused_feed_forward")
    elif "cross_block_kernel" in name:
        out.append("roofline_fused_cross_attention")
    elif "temporal_attention_kernel" in name or "temporal_stream_kernel" in name:
        out.append("roofline_temporal")
    elif "attention_kernel" in name or "attention_dma_kernel" in name:
        out.append("attention")
    return out


acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        for key in classes(row["Kernel_Name"]):
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
out, detail = {}, {}
for key, cs in acc.items():
    d = {}
    fetch, write = cs.get("FETCH_SIZE", []), cs.get("WRITE_SIZE", [])
    if fetch and write:
        fb = 2.0 * 1024.0 * sum(fetch) / len(fetch)
        wb = 1024.0 * sum(write) / len(write)
        out[key] = fb + wb
        d.update(launches_fetch_pass=len(fetch), launches_write_pass=len(write), fetch_bytes_per_launch_x2_corrected=fb,
                 write_bytes_per_launch=wb)
    busy, act = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", []), cs.get("GRBM_GUI_ACTIVE", [])
    if busy and act and sum(act) > 0:
        d.update(launches_mfma_pass=len(busy), mfma_busy_cycles_per_launch=sum(busy) / len(busy),
                 gui_active_per_launch_sum_over_8_xcds=sum(act) / len(act),
                 mfma_busy_fraction_at_held_clock=sum(busy) / ((sum(act) / 8.0) * 1024.0))
    if d:
        detail[key] = d
out["_detail"] = detail
out["_note"] = ("HBM-side bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction), averaged over every launch "
                "of the class in a 2-forward bench run; mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024)")
print(json.dumps(out, indent=1))

#!/bin/bash
# Temporal-attention LDS budget sweep on the 61-frame interpolation forward (tuning; run through gpurun).
for b in 33000 70000 135000; do
  LAVIE_TEMPORAL_BUDGET=$b timeout -k 10 200 python3 tools/bench_interp.py --steps 6 2>/dev/null > gpurun_out/tb_$b.json
  python3 - "$b" <<'PY'
import json, sys
b = sys.argv[1]
d = json.load(open(f"gpurun_out/tb_{b}.json"))
print(b, round(d["ms_per_unet_forward"], 2), [c for c in d["kernel_breakdown"] if c["name"] == "temporal_attention"])
PY
done

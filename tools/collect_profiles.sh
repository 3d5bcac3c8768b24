#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, its rocprofv3 kernel trace, and the PMC passes behind
# roofline.traffic / MFMA busy.  Outputs under gpurun_out/; copy the summaries into profiles/ afterwards.
# The traced and counter runs pass --headline-only: only the headline configuration runs (one prompt per forward, no extra legs), so that
# per-kernel averages and bytes per launch are those of the launches bench.py's `roofline` objects count.
# Usage: bash tools/collect_profiles.sh <tag>
set -o pipefail
TAG=${1:-r01_d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$PMC_ONLY" ]; then
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done: $(head -c 200 $OUT/bench.json)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 1 --cpu-steps 0 --headline-only > $OUT/bench_traced.json 2> $OUT/trace.err || exit 1
python3 tools/summarize_trace.py $OUT/trace > $OUT/kernel_summary.md
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
echo "trace done"
fi
mkdir -p $OUT/pmc
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    D=$OUT/pmc/$(echo $C | tr ' ' '_')
    rocprofv3 --pmc $C --output-format csv -d $D -- python3 bench.py --steps 1 --warmup 0 --ddpm-steps 2 --cpu-steps 0 --no-profile --headline-only > $D.json 2> $D.err || echo "pmc pass $C failed"
    echo "pmc $C done"
done
python3 tools/pmc_traffic.py $OUT/pmc > $OUT/pmc_traffic.json
rm -rf $OUT/trace $OUT/pmc/*/
head -c 600 $OUT/pmc_traffic.json

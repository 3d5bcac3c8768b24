#!/bin/bash
# Development aid: PMC passes over a few launches of the level-0 self-attention (tools/pmc_probe.py attn).  Stops at the first pass that
# is killed by its timeout; a pass that rocprofv3 refuses (unknown counter) is reported and skipped.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_attn
mkdir -p $OUT
rocprofv3 -L > $OUT/avail.txt 2>&1
i=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
         "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_MFMA" \
         "SQ_INSTS_VALU_TRANS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
         "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 tools/pmc_probe.py attn > $OUT/p$i.log 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i killed by timeout: stop"; exit 1; fi
    if [ $rc -ne 0 ]; then echo "pass $i ($C) refused: rc $rc: $(tail -2 $OUT/p$i.log | head -1)"; continue; fi
    f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
    python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "attention_dma_kernel" not in r["Kernel_Name"]:
        continue
    a = acc[r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"  {k:34s} per launch {v / n:16.0f}   ({n} rows)")
PY
    rm -rf $OUT/p$i
done

#!/bin/bash
# LDS counters of the training lm_head GEMM (one pass, SQ block only): bank conflicts, LDS-array cycles, LDS instructions
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_gemm_lds
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 tools/prof_gemm.py > $OUT/p1.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_gemm_lds/p1/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gemm_nt8_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:28s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
PY

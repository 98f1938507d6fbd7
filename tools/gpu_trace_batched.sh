#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace of the batch-32 leg and its per-cycle summary + timeline.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$1
mkdir -p $OUT
timeout -k 10 200 python3 tools/batched_only.py > $OUT/batched.json 2> $OUT/batched.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_batched -- python3 tools/batched_only.py > $OUT/batched_trace.log 2>&1
python3 tools/busy_summary.py $OUT/trace_batched 6 1 > $OUT/batch32_cycle_timeline.txt
cat $OUT/batched.json | cut -c1-200
head -24 $OUT/batch32_cycle_timeline.txt

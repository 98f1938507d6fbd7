#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace of the decode bench and the timeline of one steady-state cycle.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$1
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_decode -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-train --no-batched > $OUT/bench_trace.log 2>&1
python3 tools/cycle_timeline.py $OUT/trace_decode > $OUT/cycle_timeline.txt
tail -14 $OUT/cycle_timeline.txt

#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats of the bench, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$1
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --train-steps 4 --train-warmup 1 > $OUT/bench_trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_lm_head.py > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/prof_lm_head.py > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20

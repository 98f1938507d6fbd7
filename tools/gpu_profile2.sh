#!/bin/bash
# Runs on the GPU box (through gpurun): decode-cycle kernel trace, then PMC passes for the fused loss kernels
# (FETCH_SIZE / WRITE_SIZE, separate passes) and for the training lm_head GEMM (MFMA busy cycles).
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$1
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_decode -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-train --no-batched > $OUT/bench_trace.log 2>&1
python3 tools/cycle_timeline.py $OUT/trace_decode > $OUT/cycle_timeline.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_loss_fetch -- python3 tools/prof_loss.py > $OUT/pmc_loss_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_loss_write -- python3 tools/prof_loss.py > $OUT/pmc_loss_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_gemm -- python3 tools/prof_gemm.py > $OUT/pmc_gemm.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_gemm -- python3 tools/prof_gemm.py > $OUT/trace_gemm.log 2>&1
find $OUT -name "*.csv" | head -20
tail -30 $OUT/cycle_timeline.txt

#!/bin/bash
# Runs on the GPU box (through gpurun): every artifact profiles/r03_* is made from.  One argument: a tag for gpurun_out/prof_<tag>.
#   1. the default bench line (no profiler attached)                       -> bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command (short)        -> trace/  (kernel stats, lm_head trace rows)
#   3. decode cycle timeline, batch-32 cycle timeline, training step timeline
#   4. PMC passes of the dominant kernel (FETCH_SIZE, WRITE_SIZE: separate passes)
# Each GPU step has its own time limit; the script stops at the first failure.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$1
mkdir -p $OUT
timeout -k 10 420 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --train-steps 3 --train-warmup 1 > $OUT/bench_trace.log 2>&1
python3 tools/cycle_timeline.py $OUT/trace > $OUT/decode_cycle_timeline.txt
python3 tools/train_timeline.py $OUT/trace > $OUT/train_step_timeline.txt
python3 tools/trace_summary.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) "keepx_kernel<1, 0, 4" $OUT/lm_head_trace.json > /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_batched -- python3 tools/batched_only.py > $OUT/batched_trace.log 2>&1
python3 tools/busy_summary.py $OUT/trace_batched 6 1 > $OUT/batch32_cycle_timeline.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_lm_head.py > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/prof_lm_head.py > $OUT/pmc_write.log 2>&1
python3 tools/pmc_to_json.py $OUT $OUT/lm_head_pmc.json > /dev/null
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
# fused loss kernels (HBM traffic) and the training lm_head GEMM (MFMA pipe occupancy): refreshed every round
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_loss_fetch -- python3 tools/prof_loss.py > $OUT/pmc_loss_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_loss_write -- python3 tools/prof_loss.py > $OUT/pmc_loss_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_gemm -- python3 tools/prof_gemm.py > $OUT/pmc_gemm.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_gemm -- python3 tools/prof_gemm.py > $OUT/trace_gemm.log 2>&1
python3 tools/pmc2_to_json.py $OUT $OUT/r03 > /dev/null
tail -c 600 $OUT/bench.json

#!/bin/bash
# Runs on the GPU box: the training leg alone (tokens/s) and its per-kernel totals from a kernel trace.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$1
mkdir -p $OUT
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-batched --steps 20 --warmup 3 > $OUT/bench_train.json 2> $OUT/bench_train.err
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_train.json").read().strip().splitlines()[-1])
print("decode", round(d["value"],1), "train", round(d["train"]["value"],0), "ms/step", round(d["train"]["ms_per_step"],2), "frac", round(d["train"]["roofline"]["frac"],3))
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -- python3 bench.py --no-cpu-baseline --no-batched --steps 5 --warmup 1 --train-steps 3 --train-warmup 1 > $OUT/train_trace.log 2>&1
python3 tools/train_timeline.py $OUT/trace_train > $OUT/train_step_timeline.txt
head -30 $OUT/train_step_timeline.txt

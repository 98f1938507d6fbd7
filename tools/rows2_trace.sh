#!/bin/bash
# GPU box: kernel trace of tools/rows2_bench.py, per (kernel, grid) average durations.  Arguments: the bench's knobs.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/rows2_trace
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 tools/rows2_bench.py "$@" > $OUT/bench.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_rows" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
runs = []          # consecutive launches of the same (kernel, grid): one bench shape each
for r in rows:
    n = r["Kernel_Name"]
    k = (n[:60], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    stream = "stream" in n
    if stream:
        if runs and runs[-1][0] == k: runs[-1][1].append(d)
        else: runs.append([k, [d], []])
    else:
        runs[-1][2].append((k, d))
for k, v, red in runs:
    v = sorted(v)
    line = f"{v[len(v)//2]:8.1f} us  x{len(v):3d}  wg={k[1]:5d}  {k[0][5:50]}"
    if red:
        rd = sorted(d for _, d in red)
        line += f"   + reduce {rd[len(rd)//2]:5.1f} us (grid {red[0][0][1]} x {red[0][0][2]})"
    print(line)
PY
grep "M=" $OUT/bench.txt

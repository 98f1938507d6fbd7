#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_r03
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 100 python3 tools/prof_lm_head_variants.py > $OUT/timing.txt 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $OUT/p1 -- python3 tools/prof_lm_head_variants.py > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum TCC_READ_sum --output-format csv -d $OUT/p2 -- python3 tools/prof_lm_head_variants.py > $OUT/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 tools/prof_lm_head_variants.py > $OUT/p3.log 2>&1
python3 tools/pmc_variants_table.py $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/table.txt
cat $OUT/timing.txt | grep variant; cat $OUT/table.txt

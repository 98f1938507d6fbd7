"""Turn the two rocprofv3 --pmc passes of tools/gpu_profile.sh (FETCH_SIZE, WRITE_SIZE of
tools/prof_lm_head.py) into profiles/<name>.json, the file bench.py's `roofline.traffic` reads.
    python tools/pmc_to_json.py gpurun_out/prof_X profiles/r02_lm_head_pmc.json
"""
import csv
import glob
import json
import sys

KERNEL = "gemm_skinny_keepx_kernel"
V, D, ROWS = 32000, 4096, 4


def per_launch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter]


root, out = sys.argv[1], sys.argv[2]
fetch = per_launch(root + "/pmc_fetch", "FETCH_SIZE")
write = per_launch(root + "/pmc_write", "WRITE_SIZE")
# gfx950: FETCH_SIZE counts a 128-B request as 64 B for wide coalesced reads -> x2; WRITE_SIZE exact; KiB
traffic = (sum(fetch) / len(fetch) * 2 + sum(write) / len(write)) * 1024
BLOCKS, TOPK = 250, 10
alg = V * D * 2 + ROWS * D * 2 + ROWS * BLOCKS * TOPK * 8       # weights + x + candidate records (no logits leave the kernel)
json.dump({
    "kernel": "gemm_skinny_keepx_kernel<1,0,4,...,packed,TKP=4> (lm_head, %d rows x %d x %d bf16, packed weight image, top-k candidate epilogue)" % (ROWS, V, D),
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 tools/prof_lm_head.py (separate passes)",
    "FETCH_SIZE_KB_per_launch": fetch,
    "WRITE_SIZE_KB_per_launch": write,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; unit KiB.  "
                  "Round 3 (profiles/r03_lm_head_fetch_variants.txt): FETCH_SIZE = TCC_EA0_RDREQ x 64 B whatever the request size, so x2 is right only when every "
                  "memory request is a whole line, as with the packed image.  The row-major fragment shape (non-temporal half-line pairs) sends 8.3 % more memory "
                  "requests than lines -- lines fetched twice or as two halves, these counters cannot tell -- which read as 1.085 x in round 2",
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": traffic / alg,
}, open(out, "w"), indent=1)
print(open(out).read())

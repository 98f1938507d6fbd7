"""rocprofv3 --pmc CSVs of tools/prof_lm_head_variants.py -> one table: per variant (dispatch order) the mean of every counter."""
import csv
import glob
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_skinny_keepx_kernel" in k or "reduce_kernel" in k:
                per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), k, float(r["Counter_Value"])))
        for c, lst in per.items():
            lst.sort()
            gem = [v for _, k, v in lst if "keepx" in k]
            red = [v for _, k, v in lst if "reduce_kernel" in k]
            for v in range(5):
                rows[v][c] += gem[3 * v:3 * v + 3]
            rows[5][c] += red
            rows[6][c] += gem[15:18]          # the packed image (dispatched last)
names = ["RT1 rot nt (default)", "RT2 rot nt", "RT1 norot nt", "RT2 norot nt (r01)", "RT1 rot default-policy", "torch sum 262 MB", "packed image nt (shipped)"]
cs = sorted({c for v in rows.values() for c in v})
print("%-26s" % "variant" + "".join("%24s" % c for c in cs))
for v in range(7):
    print("%-26s" % names[v] + "".join("%24.1f" % (sum(rows[v][c]) / max(1, len(rows[v][c]))) for c in cs))

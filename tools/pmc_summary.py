"""Average the rocprofv3 --pmc counters of the kernels whose name contains argv[2]."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")

"""Average of each PMC counter per launch for kernels whose name contains argv[2], over the last argv[3] launches."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
pat, last = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    v = v[-last:]
    print(f"{k:28s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")

"""Per-shape breakdown of one kernel in a rocprofv3 --kernel-trace csv: the same instantiation serves
several GEMM shapes (lm_head and qkv both run gemm_skinny_keepx_kernel<2,0,4>), and --stats averages
them together; the launch grid tells them apart.
    python tools/trace_summary.py <kernel_trace.csv> "keepx_kernel<2, 0, 4>" profiles/out.json
"""
import collections
import csv
import json
import sys

rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg[(r["Kernel_Name"], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_X"]))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = [{"kernel": k[0], "workgroups": k[1], "workgroup_size": k[2], "calls": len(v), "avg_ns": sum(v) / len(v),
        "min_ns": min(v), "max_ns": max(v)} for k, v in sorted(agg.items())]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))

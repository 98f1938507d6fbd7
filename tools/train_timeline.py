"""Aggregate one steady-state training step (between two adamw launches) from a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adamw')]
i0, i1 = ad[-2] + 1, ad[-1] + 1
t0 = int(rows[i0]['Start_Timestamp'])
agg, gap, prev_end = {}, 0, t0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap += max(0, s - prev_end); prev_end = max(prev_end, e)
    n = r['Kernel_Name'][:50]
    k = agg.setdefault(n, [0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3
print("step span us", (int(rows[i1 - 1]['End_Timestamp']) - t0) / 1e3, "idle gaps us", gap / 1e3)
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{t:9.1f} us x{c:3d} {n}")
if len(sys.argv) > 2:
    for r in rows[i0:i1]:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} {r['Kernel_Name'][:40]}")

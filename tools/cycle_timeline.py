"""Print the kernel timeline of one steady-state draft/verify cycle from a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('greedy_accept')]
i0 = idx[-30] if len(idx) > 30 else idx[len(idx) // 2]
i1 = idx[idx.index(i0) + 1]
t0 = int(rows[i0]['Start_Timestamp'])
agg = {}
for r in rows[i0:i1]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    name = r['Kernel_Name'][:48]
    print(f"{s/1e3:8.1f} dur={(e-s)/1e3:6.2f} grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>6d}x{r['Grid_Size_Y']:>3s} wg={r['Workgroup_Size_X']:>4s} vgpr={r['VGPR_Count']:>3s} {name}")
    k = agg.setdefault(name, [0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3
print("cycle span us:", (int(rows[i1]['Start_Timestamp']) - t0) / 1e3)
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:8.1f} us  x{c:3d}  {n}")

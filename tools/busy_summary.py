"""GPU-busy fraction and per-kernel totals per draft/verify cycle over the last `n` cycles of a rocprofv3 kernel trace
(cycles are cut at greedy_accept launches)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('greedy_accept')]
i0, i1 = idx[-n - 1], idx[-1]
sel = rows[i0:i1]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel)
span = int(rows[i1]['Start_Timestamp']) - int(sel[0]['Start_Timestamp'])
print(f"{n} cycles: {span/n/1e3:.1f} us per cycle, {len(sel)/n:.0f} kernels per cycle, busy {busy/n/1e3:.1f} us = {busy/span:.2%}")
agg = {}
for r in sel:
    k = agg.setdefault(r['Kernel_Name'][:60], [0, 0]); k[0] += 1; k[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/n/1e3:10.1f} us/cycle x{c/n:6.1f} avg {t/c/1e3:8.1f}  {nm}")
if len(sys.argv) > 3:
    t0 = int(rows[idx[-2]]['Start_Timestamp'])
    for r in rows[idx[-2]:idx[-1]]:
        s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
        print(f"{s/1e3:8.1f} dur={(e-s)/1e3:7.2f} grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>6d}x{r['Grid_Size_Y']:>3s}x{r['Grid_Size_Z']:>3s} {r['Kernel_Name'][:50]}")

import csv, sys, glob
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
lo = int(sys.argv[2]) if len(sys.argv) > 2 else n - 400
hi = int(sys.argv[3]) if len(sys.argv) > 3 else lo + 75
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print(f"{s/1e3:9.1f} dur={(e-s)/1e3:7.2f} grid={r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:>3s} wg={r['Workgroup_Size_X']:>4s} vgpr={r['VGPR_Count']:>4s} {r['Kernel_Name'][:70]}")

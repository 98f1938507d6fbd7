"""Print the kernel timeline of one steady-state draft/verify cycle from a rocprofv3 kernel trace.
A cycle = the launches from one greedy_accept to the next.  The trace of bench.py holds graph-replayed cycles (the timed region),
eager cycles (warm-up, the roofline probe, which skips the merge launch) and the other legs' cycles: the cycle printed is the
one with the SHORTEST span among those whose kernel sequence is the most common one (= a graph replay of the timed region)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('greedy_accept')]
cyc = []
for a, b in zip(idx[:-1], idx[1:]):
    sig = tuple((r['Kernel_Name'], r['Grid_Size_X'], r['Grid_Size_Y']) for r in rows[a:b])
    cyc.append((sig, int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']), a, b))
common = collections.Counter(c[0] for c in cyc).most_common(1)[0][0]
same = [c for c in cyc if c[0] == common]
_, span, i0, i1 = min(same, key=lambda c: c[1])
spans = sorted(c[1] for c in same)
t0 = int(rows[i0]['Start_Timestamp'])
agg = {}
for r in rows[i0:i1]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    name = r['Kernel_Name'][:48]
    print(f"{s/1e3:8.1f} dur={(e-s)/1e3:6.2f} grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>6d}x{r['Grid_Size_Y']:>3s} wg={r['Workgroup_Size_X']:>4s} vgpr={r['VGPR_Count']:>3s} {name}")
    k = agg.setdefault(name, [0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3
print(f"cycle span us: {span/1e3:.1f}   ({len(same)} cycles with this kernel sequence: min {spans[0]/1e3:.1f}, median {spans[len(spans)//2]/1e3:.1f} us; under the profiler)")
print(f"sum of kernel durations us: {sum(t for _, t in agg.values()):.1f}   launches: {sum(c for c, _ in agg.values())}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:8.1f} us  x{c:3d}  {n}")

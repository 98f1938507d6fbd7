import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
k = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'keepx' in r['Kernel_Name']]
reps = 12
for i, name in enumerate(("no touch", "64 KiB touched", "256 KiB touched")):
    seg = k[i * 2 * reps:(i + 1) * 2 * reps]
    ev, lm = sorted(seg[0::2]), sorted(seg[1::2])
    print(f"{name:16s} evicting stream median {ev[len(ev)//2]:.1f} us   lm_head median {lm[len(lm)//2]:.1f} us (min {lm[0]:.1f})")

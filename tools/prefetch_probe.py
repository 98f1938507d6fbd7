"""Does a warm first stage shorten a weight-stream launch?  (round 3 probe for the 'tail of the previous launch touches the next
matrix' idea.)  Per repetition: a 262 MB stream over ANOTHER matrix (evicts), optionally a torch reduction over the first
`touch_kb` KiB of every block's 1 MiB region of the packed lm_head image (default-policy loads: lines stay in L2 / Infinity Cache),
then the lm_head launch.  Run under rocprofv3 --kernel-trace; the per-kernel durations of the keepx launches are printed by
tools/prefetch_probe_table.py.  Order of keepx dispatches: [evict, lm_head] x reps without touch, then with 64 KiB, then with 256 KiB."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
V, D, rows, reps = 32000, 4096, 4, 12
W = (torch.randn((V, D), generator=g, device=dev) * 0.02).to(torch.bfloat16)
W2 = (torch.randn((V, D), generator=g, device=dev) * 0.02).to(torch.bfloat16)
x = torch.randn((rows, D), generator=g, device=dev).to(torch.bfloat16)
Wp, W2p = ops.pack_weight(W), ops.pack_weight(W2)
out = torch.empty((rows, V), device=dev, dtype=torch.bfloat16)
img = Wp.data.view(torch.int32).view(250, -1)          # one row per block of the 250-block launch: 1 MiB each
ev = {}
for touch_kb in (0, 64, 256):
    torch.cuda.synchronize()
    ts = []
    for r in range(reps):
        ops.linear_skinny(x, W2p, out=out)
        if touch_kb:
            s = img[:, : touch_kb * 256].sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.linear_skinny(x, Wp, out=out); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
    print(f"touch {touch_kb:3d} KiB per block: lm_head launch median {v[len(v)//2]:.1f} us (min {v[0]:.1f}) by events", flush=True)

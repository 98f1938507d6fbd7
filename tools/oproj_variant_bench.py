"""o_proj-sized (4096 x 4096) and down-sized GEMMs over packed weights, 4 rows: tile kernel vs stream vs register-resident x (cold caches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
from eagle_amd._hip import lib
BF = torch.bfloat16
d = torch.device("cuda:0")
g = torch.Generator(device=d).manual_seed(0)
rnd = lambda *s: (torch.randn(s, generator=g, device=d, dtype=torch.float32) * 0.02).to(BF)
scrub = rnd(16384, 8192)
for (N, K, name) in ((4096, 4096, "o_proj"), (4096, 11008, "down"), (4096, 8192, "fc")):
    W = ops.pack_weight(rnd(N, K)); x = rnd(4, K) * 30; r = rnd(4, N)
    for var, vn in ((3, "default"), (0, "tile kernel"), (1, "stream kernel"), (2, "keepx where legal")):
        lib().eagle_debug_set_skinny_variant(var, 0, 0)
        ts = []
        for i in range(12):
            ops.linear_skinny(scrub[:4], scrub)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.linear_skinny(x, W, res=r); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"{name:7s} {vn:18s}: median {ts[6]:6.2f} us  min {ts[0]:6.2f}", flush=True)
lib().eagle_debug_set_skinny_variant(3, 0, 0)

"""Row-major vs packed weights on the decode GEMMs of one A-dims draft forward (4 rows): outputs must be bit-equal; time per
launch from cold caches (a different matrix streams in between) and the sum over a forward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops  # noqa: E402

BF = torch.bfloat16
d = torch.device("cuda:0")
g = torch.Generator(device=d).manual_seed(0)


def rnd(*s):
    return (torch.randn(s, generator=g, device=d, dtype=torch.float32) * 0.02).to(BF)


M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
D, I, V = 4096, 11008, 32000
shapes = {"fc": (D, 2 * D), "qkv": (3 * D, D), "o": (D, D), "gate|up(silu)": (I, D), "down": (D, I), "lm_head": (V, D)}
scrub = rnd(16384, 8192)                                   # 268 MB: streams between timed launches
total = {False: 0.0, True: 0.0}
for name, (N, K) in shapes.items():
    W = rnd(N, K)
    W2 = rnd(N, K) if "silu" in name else None
    x = rnd(M, K) * 30
    Wp, W2p = ops.pack_weight(W), (ops.pack_weight(W2) if W2 is not None else None)
    kw = {"norm_eps": 1e-6} if W2 is not None else {}
    y0 = ops.linear_skinny(x, W, W2=W2, **kw)
    y1 = ops.linear_skinny(x, Wp, W2=W2p, **kw)
    assert torch.equal(y0, y1), name
    res = {}
    for packed in (False, True, False, True):
        ts = []
        for i in range(8):
            ops.linear_skinny(scrub[:4], scrub)                # something else streams first (cold caches, dependent-launch shape)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.linear_skinny(x, Wp if packed else W, W2=(W2p if packed else W2), **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[packed] = ts[len(ts) // 2]
    mb = N * K * 2 * (2 if W2 is not None else 1) / 1e6
    total[False] += res[False]; total[True] += res[True]
    print(f"{name:14s} {mb:7.1f} MB  row-major {res[False]:6.1f} us ({mb / res[False]:5.2f} TB/s)   packed {res[True]:6.1f} us ({mb / res[True]:5.2f} TB/s)", flush=True)
print(f"sum over one forward: row-major {total[False]:.1f} us, packed {total[True]:.1f} us")

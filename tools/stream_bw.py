"""Calibration: skinny-GEMM streaming rate vs weight size (beyond the 256 MiB Infinity Cache) and a
torch copy of the same bytes for reference."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
import ctypes
from eagle_amd import _hip
setv = _hip.lib().eagle_debug_set_skinny_variant
shapes = [(4096, 4096), (12288, 4096), (32000, 4096), (151936, 4096)]
for M in (1, 8):
    for N, K in shapes:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        Ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(max(1, min(8, int(6e8 // (N * K * 2)))))]
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res = []
        for var, bpc, mr in ((0, 2, 16), (1, 2, 16), (2, 2, 16)):
            setv(var, bpc, mr)
            it = [0]
            def f():
                it[0] = (it[0] + 1) % len(Ws)
                ops.linear_skinny(x, Ws[it[0]], out=o)
            s = t(f, 24)
            res.append(f"v{var}/b{bpc}/r{mr}: {s*1e6:7.1f} us {N*K*2/s/1e12:5.2f} TB/s")
        print(f"M={M:2d} N={N:6d} K={K:5d}  " + " | ".join(res), flush=True)
        del Ws
setv(1, 2, 8)
for M in ():
    x = torch.randn(M, 4096, device=dev).to(torch.bfloat16)
    for N in (4096, 12288, 32000, 151936):
        W = (torch.randn(N, 4096, device=dev) * 0.02).to(torch.bfloat16)
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        s = t(lambda: ops.linear_skinny(x, W, out=o))
        print(f"M={M:2d} N={N:6d} K=4096: {s*1e6:8.1f} us  {N*4096*2/s/1e12:5.2f} TB/s", flush=True)
        del W
for nbytes in (262144000, 1244659712):
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    s = t(lambda: b.copy_(a))
    print(f"torch copy {nbytes/1e6:.0f} MB: {s*1e6:8.1f} us  read+write {2*nbytes/s/1e12:5.2f} TB/s", flush=True)
    s = t(lambda: a.sum())
    print(f"torch sum  {nbytes/1e6:.0f} MB: {s*1e6:8.1f} us  read {nbytes/s/1e12:5.2f} TB/s", flush=True)

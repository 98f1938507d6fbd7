"""Calibration: period of trivial / small kernels inside a torch-captured HIP graph on this box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device("cuda:0")
flat = torch.arange(1000, device=dev)
idx = torch.arange(8, device=dev)
out = torch.empty((1, 8), dtype=torch.int64, device=dev)
x = torch.randn(8, 4096, device=dev).to(torch.bfloat16)
w = torch.ones(4096, device=dev, dtype=torch.bfloat16)

def bench(fn, n, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print(f"{name:40s} graph of {n}: {dt*1e6/n:7.2f} us per kernel", flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{'':40s} eager   {n}: {e0.elapsed_time(e1)*1e3/n:7.2f} us per kernel", flush=True)

bench(lambda: ops.tree_gather_ids(flat, idx, out=out), 200, "tree_gather_ids (1 wave)")
bench(lambda: ops.rmsnorm(x, w, 1e-6), 200, "rmsnorm 8x4096 (+alloc)")
W = (torch.randn(4096, 4096, device=dev) * 0.02).to(torch.bfloat16)
o = torch.empty(8, 4096, device=dev, dtype=torch.bfloat16)
bench(lambda: ops.linear_skinny(x, W, out=o), 100, "skinny 8x4096x4096 (33.5MB)")
Ws = [(torch.randn(4096, 4096, device=dev) * 0.02).to(torch.bfloat16) for _ in range(16)]
it = [0]
def rot():
    it[0] = (it[0] + 1) % 16
    ops.linear_skinny(x, Ws[it[0]], out=o)
bench(rot, 96, "skinny 4096x4096 rotating 16 weights")
W2 = (torch.randn(32000, 4096, device=dev) * 0.02).to(torch.bfloat16)
o2 = torch.empty(8, 32000, device=dev, dtype=torch.bfloat16)
def mixed():
    rot(); ops.linear_skinny(x, W2, out=o2)
bench(mixed, 48, "pair: skinny 4096^2 + lm_head 32000x4096")

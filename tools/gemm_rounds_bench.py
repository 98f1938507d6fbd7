"""lm_head forward of the training step: one launch over 9504 tiles (37 rounds + 32 tiles) vs whole rounds + split-K tail (ops.gemm_nt_rounds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device("cuda:0")
import sys as _s
M, N, K = (4096, 151936, 4096) if len(_s.argv) < 4 else tuple(int(v) for v in _s.argv[1:4])
print("shape", M, N, K, "plan", ops.gemm_nt_tail_plan(M, N, K))
A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
res = {"one launch": [], "rounds + tail": []}
for rnd in range(5):
    for which, fn in (("one launch", lambda: ops.gemm_nt(A, B, out=C)), ("rounds + tail", lambda: ops.gemm_nt_rounds(A, B, out=C))):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        res[which].append(e0.elapsed_time(e1) / 5)
for which, v in res.items():
    v.sort()
    print(f"{which:14s}: median {v[len(v)//2]*1e3:8.1f} us = {2*M*N*K/v[len(v)//2]/1e9:7.1f} TFLOP/s   (min {v[0]*1e3:8.1f} us)", flush=True)

"""lm_head dgrad (4096 x 4096 x 151936, 256 output tiles): split-K factor sweep."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device("cuda:0")
M, N, K = 4096, 4096, 151936
A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
res = {ks: [] for ks in (1, 2, 3, 4, 8)}
for rnd in range(4):
    for ks in res:
        fn = lambda: ops.gemm_nt(A, B, out=C, ksplit=ks)
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        res[ks].append(e0.elapsed_time(e1) / 5)
for ks, v in res.items():
    v.sort()
    print(f"ksplit {ks}: median {v[len(v)//2]*1e3:8.1f} us = {2*M*N*K/v[len(v)//2]/1e9:7.1f} TFLOP/s (GEMM + slab sum)", flush=True)

"""lm_head forward on target and predict: two 4096-row GEMMs vs one 8192-row GEMM over the stacked rows (same weights)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device("cuda:0")
M, N, K = 4096, 151936, 4096
A2 = torch.randn(2 * M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
C2 = torch.empty(2 * M, N, device=dev, dtype=torch.bfloat16)
print("plans", ops.gemm_nt_tail_plan(M, N, K), ops.gemm_nt_tail_plan(2 * M, N, K))
def two():
    ops.gemm_nt_rounds(A2[:M], B, out=C2[:M]); ops.gemm_nt_rounds(A2[M:], B, out=C2[M:])
def one():
    ops.gemm_nt_rounds(A2, B, out=C2)
res = {"two launches of 4096 rows": [], "one launch of 8192 rows": []}
for rnd in range(4):
    for which, fn in (("two launches of 4096 rows", two), ("one launch of 8192 rows", one)):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        res[which].append(e0.elapsed_time(e1) / 3)
for which, v in res.items():
    v.sort()
    print(f"{which:28s}: median {v[len(v)//2]*1e3:8.1f} us = {2*2*M*N*K/v[len(v)//2]/1e9:7.1f} TFLOP/s", flush=True)

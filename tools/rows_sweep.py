"""Decode-batch lm_head GEMM (M rows x 151936 x 4096) by kernel configuration, from cold caches.  GPU box only."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eagle_amd import _hip, ops
dev = torch.device("cuda:0")
setm = _hip.lib().eagle_debug_set_gemm8
N, K = 151936, 4096
Ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(2)]
for M in (32, 128, 192):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ref = None
    for name, cfg in (("128x128 (old)", 0x1fff), ("rows BM32 S3", 0x123), ("rows BM32 S4", 0x124), ("rows BM64 S3", 0x143),
                      ("rows BM128 S3", 0x183), ("128x256 S3 8w", 0x193)):
        setm(cfg)
        out = ops.gemm_nt(x, Ws[0])
        if ref is None: ref = out.clone()
        err = float((out.float() - ref.float()).abs().max())
        for i in range(4): ops.gemm_nt(x, Ws[i % 2])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10): ops.gemm_nt(x, Ws[i % 2])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"M={M:4d} {name:16s} {us:8.1f} us  {N*K*2/us/1e6:5.2f} TB/s  maxdiff vs old {err:.3g}", flush=True)
setm(0x100)

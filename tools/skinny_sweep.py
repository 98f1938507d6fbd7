"""Weight-stream GEMM variants from cold caches (rotating weight copies), us per launch and TB/s.  GPU box only."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eagle_amd import _hip, ops
dev = torch.device("cuda:0")
setv = _hip.lib().eagle_debug_set_skinny_variant
shapes = [("lm_head", 32000, 4096, False), ("qkv", 12288, 4096, False), ("o", 4096, 4096, False), ("gate|up", 11008, 4096, True),
          ("lm_head qwen", 151936, 4096, False)]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
variants = [("default", 3), ("keepx S2 rt2", 2 | (2 << 4) | (1 << 8) | (2 << 12)), ("keepx S4 rt2", 2 | (4 << 4) | (1 << 8) | (2 << 12)),
            ("keepx S2 rt1", 2 | (2 << 4) | (1 << 8) | (1 << 12)), ("keepx S4 rt1", 2 | (4 << 4) | (1 << 8) | (1 << 12)),
            ("keepx S2 rt2 bpc2", 2 | (2 << 4) | (2 << 8) | (2 << 12)), ("keepx S2 rt1 bpc2", 2 | (2 << 4) | (2 << 8) | (1 << 12)),
            ("keepx S4 rt1 bpc2", 2 | (4 << 4) | (2 << 8) | (1 << 12)), ("stream", 1), ("tile", 0)]
for name, N, K, silu in shapes:
    ncopy = max(2, int(1.2e9 // (N * K * 2)) + 1)
    Ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(ncopy)]
    W2 = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) if silu else None
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for vname, v in variants:
        setv(v, 0, 0)
        try:
            for i in range(ncopy): ops.linear_skinny(x, Ws[i], W2=W2, out=out, norm_eps=1e-6 if silu else 0.0)
            torch.cuda.synchronize()
            it = 5 * ncopy
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(it): ops.linear_skinny(x, Ws[i % ncopy], W2=W2, out=out, norm_eps=1e-6 if silu else 0.0)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / it * 1e3
            nbytes = N * K * 2 * (2 if silu else 1)
            print(f"{name:14s} M={M} {vname:20s} {us:8.1f} us  {nbytes/us/1e6:6.2f} TB/s", flush=True)
        except Exception as e:
            print(name, vname, "failed", e, flush=True)
    setv(3 | (2 << 4) | (1 << 8), 0, 0)
    del Ws

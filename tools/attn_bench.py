"""Training attention fwd/bwd micro-benchmark at the bench shape (B=1, L=2048, H=32, hd=128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device("cuda:0")
B, L, H, hd = 1, 2048, 32, 128
qkv = (torch.randn(B * L, 3 * H * hd, device=dev) * 0.5).to(torch.bfloat16)
dout = torch.randn(B * L, H * hd, device=dev).to(torch.bfloat16)
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out, lse = ops.attn_train_fwd(qkv, B, L, H, hd)
fl = 4.0 * B * H * L * L / 2 * hd
us = t(lambda: ops.attn_train_fwd(qkv, B, L, H, hd))
print(f"fwd {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s (causal flops)")
us = t(lambda: ops.attn_train_bwd(qkv, out, dout, lse, B, L, H, hd))
print(f"bwd {us:8.1f} us  {2.5*fl/us/1e6:7.1f} TFLOP/s")

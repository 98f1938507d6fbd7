"""Calibration of the box ('take peak numbers from the box'): HBM stream rates of plain torch kernels
over 2 GiB buffers (far beyond the 256 MiB Infinity Cache) and of this library's weight-stream kernel on the largest
matrix of the hot path, so that roofline fractions can also be read against what the chip sustains."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device("cuda:0")

def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

n = 1 << 29                                        # 2 GiB of fp32
x = torch.ones(n, dtype=torch.float32, device=dev); y = torch.empty_like(x)
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs", flush=True)
s = t(lambda: y.copy_(x)); print(f"copy  2 GiB -> 2 GiB : {2 * n * 4 / s / 1e12:5.2f} TB/s (read + write)", flush=True)
s = t(lambda: x.sum());    print(f"sum   2 GiB          : {n * 4 / s / 1e12:5.2f} TB/s (read only)", flush=True)
s = t(lambda: y.fill_(1)); print(f"fill  2 GiB          : {n * 4 / s / 1e12:5.2f} TB/s (write only)", flush=True)
del x, y
for V, name in ((151936, "qwen3 lm_head"), (32000, "llama2 lm_head")):
    Ws = [(torch.randn(V, 4096, device=dev) * 0.02).to(torch.bfloat16) for _ in range(2 if V > 100000 else 8)]
    xx = torch.randn(4, 4096, device=dev).to(torch.bfloat16); o = torch.empty(4, V, device=dev, dtype=torch.bfloat16)
    it = [0]
    def f():
        it[0] = (it[0] + 1) % len(Ws); ops.linear_skinny(xx, Ws[it[0]], out=o)
    s = t(f, 16)
    print(f"weight-stream kernel, {name} {V} x 4096 bf16, 4 rows, rotating {len(Ws)} matrices back to back: "
          f"{s * 1e6:6.1f} us = {V * 4096 * 2 / s / 1e12:5.2f} TB/s", flush=True)
    del Ws

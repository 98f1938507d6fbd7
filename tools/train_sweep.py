"""Training throughput against micro-batch size and gradient accumulation (GPU box)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eagle_amd.train_bench import train_tokens_per_s
dev = torch.device("cuda:0")
for mb, ga in ((1, 1), (2, 1), (2, 2), (4, 2), (8, 2)):
    r = train_tokens_per_s(dev, 0, 1, None, steps=4, warmup=2, ga=ga, micro_batch=mb)
    print(f"micro-batch {mb} x 2048, GA {ga}: {r['value']:.0f} tok/s, {r['ms_per_step']:.2f} ms/step, {r['roofline']['achieved']:.0f} TFLOP/s", flush=True)

"""A/B of the training step under eagle_debug_set_gemm8 words on one box: usage  python tools/train_knob_ab.py 0x30000 0x30001 ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd._hip import lib
from eagle_amd.train_bench import train_tokens_per_s
dev = torch.device("cuda:0")
words = [int(w, 0) for w in sys.argv[1:]] or [0x30000, 0x30001]
for rnd in range(2):
    for w in words:
        lib().eagle_debug_set_gemm8(w)
        r = train_tokens_per_s(dev, 0, 1, None, steps=6, warmup=2, L=2048, config="qwen3_8b", ga=2, micro_batch=2)
        print(f"knob {w:#x}: {r['ms_per_step']:.2f} ms/step  {r['value']:.0f} tok/s", flush=True)
lib().eagle_debug_set_gemm8(0x30000)

"""Fused-loss kernel bandwidth at the training shapes (GPU box)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eagle_amd.train_bench import loss_roofline
for N in (2048, 4096, 2048, 4096):
    r = loss_roofline(torch.device("cuda:0"), N, 151936, 4096)
    print(N, "fwd %.0f GB/s (%.0f us)  bwd %.0f GB/s (%.0f us)" % (r["fwd"]["GB/s"], r["fwd"]["us"], r["bwd"]["GB/s"], r["bwd"]["us"]), flush=True)

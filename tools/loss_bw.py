"""Fused-loss kernel bandwidth at the training shapes (GPU box), every load / store variant of eagle_debug_set_loss."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eagle_amd._hip import lib
from eagle_amd.train_bench import loss_roofline
names = {0: "nt loads + nt stores, 4 in flight (default)", 7: "round-2 form (default policy, 1 in flight)", 4: "nt loads, plain stores, 4 in flight",
         0x10: "nt loads + nt stores, 2 in flight", 1: "default-policy loads, nt stores, 4 in flight"}
for rnd in range(2):
    for v, name in names.items():
        lib().eagle_debug_set_loss(v)
        for N in (2048, 4096):
            r = loss_roofline(torch.device("cuda:0"), N, 151936, 4096)
            print(f"{name:46s} N={N}  fwd {r['fwd']['GB/s']:.0f} GB/s ({r['fwd']['us']:.0f} us)  bwd {r['bwd']['GB/s']:.0f} GB/s ({r['bwd']['us']:.0f} us)", flush=True)
lib().eagle_debug_set_loss(0)

"""Batch-1 decode attention (split-KV + combine) timing inside a HIP graph: fixed cost vs keys (round 3)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
BF = torch.bfloat16
d = torch.device("cuda:0")
B, H, HK, hd, cap = 1, 32, 32, 128, 2304
g = torch.Generator(device=d).manual_seed(0)
kc = torch.randn((B, HK, cap, hd), generator=g, device=d, dtype=torch.float32).to(BF)
vc = torch.randn((B, HK, cap, hd), generator=g, device=d, dtype=torch.float32).to(BF)
cos, sin = ops.rope_tables(hd, 4096, device=d)
err = torch.zeros(1, dtype=torch.int32, device=d)
for M in (4, 6):
    qkv = (torch.randn((B * M, (H + 2 * HK) * hd), generator=g, device=d, dtype=torch.float32) * 0.5).to(BF)
    out = torch.zeros((B * M, H * hd), dtype=BF, device=d)
    for kv, max_kv in ((40, 64), (40, 704), (330, 384), (330, 704), (330, 2048), (700, 704), (1500, 2048)):
        past = torch.tensor([kv], dtype=torch.int32, device=d)
        ws = torch.empty(ops.query("eagle_attn_decode_workspace_bytes", B, H, max_kv), dtype=torch.uint8, device=d)
        def call():
            ops.tree_attention(qkv, None, cos, sin, kc, vc, past, 0, None, None, out, ws, err, B, M, H, HK, hd, max_kv, True, M, 0, pos_off=0, pos_stride=1)
        call(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(50):
                call()
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"M={M} keys={kv:5d} bound={max_kv:5d}: {ts[len(ts)//2] / 50 * 1e6:6.2f} us per attention (split + combine), min {ts[0] / 50 * 1e6:6.2f}", flush=True)

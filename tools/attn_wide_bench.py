"""Wide-batch decode attention (rope/append pre-pass + attn_wide + combine) vs context length, uniform batch: us per call and
KV bytes / time.  python tools/attn_wide_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
d = torch.device("cuda:0"); BF = torch.bfloat16
B, H, HK, hd, cap = 32, 32, 32, 128, 2304
cos, sin = ops.rope_tables(hd, 4096, device=d)
kc = torch.randn(B, HK, cap, hd, device=d).to(BF); vc = torch.randn(B, HK, cap, hd, device=d).to(BF)
err = torch.zeros(1, dtype=torch.int32, device=d)
ws = torch.empty(ops.query("eagle_attn_decode_workspace_bytes", B, H, cap), dtype=torch.uint8, device=d)
scrub = torch.empty(1 << 27, dtype=torch.float32, device=d)
for M in (1, 4):
    qkv = (torch.randn(B * M, (H + 2 * HK) * hd, device=d) * 0.5).to(BF)
    out = torch.zeros((B * M, H * hd), dtype=BF, device=d)
    for ctx in (64, 256, 480, 512, 544, 1024, 2000):
        past = torch.full((B,), ctx, dtype=torch.int32, device=d)
        ts = []
        for i in range(8):
            scrub.fill_(float(i))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.tree_attention(qkv, None, cos, sin, kc, vc, past, 0, None, None, out, ws, err, B, M, H, HK, hd, ctx + M, True, M, 0, pos_off=0, pos_stride=1)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:]); us = ts[len(ts) // 2]
        mb = B * HK * (ctx + M) * hd * 2 * 2 / 1e6
        print(f"M={M} ctx={ctx:5d}  {us:7.1f} us  KV {mb:7.1f} MB  {mb / us:5.2f} TB/s", flush=True)

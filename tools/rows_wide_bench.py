"""gemm_rows_wide_kernel (up to 384-column tiles, whole rounds) against the 128-column form of eagle_gemm_rows_bf16 and the
LDS-tiled GEMM: lm_head shapes at 81..128 rows, cold caches; every variant is timed twice (the first timing after a
weight matrix is created reads ~30 us high).    python tools/rows_wide_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
from eagle_amd._hip import lib
dev = torch.device("cuda:0")
BF = torch.bfloat16
scrub = torch.empty(1 << 28, dtype=torch.float32, device=dev)


def timed(fn, n=6):
    ts = []
    for i in range(n + 2):
        scrub.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


for N, K in ((151936, 4096), (128256, 4096), (32000, 4096)):
    W = (torch.randn(N, K, device=dev) * 0.02).to(BF)
    mb = N * K * 2 / 1e6
    for M in (96, 128, 96, 128):
        x = torch.randn(M, K, device=dev).to(BF)
        lib().eagle_debug_set_rows(0, 0, 0)
        t_w = timed(lambda: ops.linear_rows(x, W))
        lib().eagle_debug_set_rows(0, 0, 0x200)
        t_n = timed(lambda: ops.linear_rows(x, W))
        lib().eagle_debug_set_rows(0, 0, 0)
        t_g = timed(lambda: ops.gemm_nt(x, W))
        d = (ops.linear_rows(x, W).float() - ops.gemm_nt(x, W).float()).abs().max().item()
        print(f"N={N} M={M:4d} {mb:7.1f} MB  wide {t_w:7.1f} us {mb / t_w:5.2f} TB/s   128-col {t_n:7.1f} us {mb / t_n:5.2f}   gemm_nt {t_g:7.1f} us {mb / t_g:5.2f}   maxdiff {d:.3g}", flush=True)

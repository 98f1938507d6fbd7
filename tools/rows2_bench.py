"""eagle_gemm_rows_bf16 against the kernels it replaces, at the batch-32 leg's shapes (Qwen3-8B dims), from cold caches.
    python tools/rows2_bench.py [target_wg min_steps stages]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops
from eagle_amd._hip import lib
dev = torch.device("cuda:0")
BF = torch.bfloat16
knobs = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else None
scrub = torch.empty(1 << 28, dtype=torch.float32, device=dev)   # 1 GiB: evicts L2 + Infinity Cache between launches


def timed(fn, n=6):
    ts = []
    for i in range(n + 2):
        scrub.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


shapes = [("fc", 4096, 8192), ("qkv", 6144, 4096), ("o", 4096, 4096), ("down", 4096, 12288), ("lm_head", 151936, 4096)]
for M in (32, 128):
    for name, N, K in shapes:
        x = torch.randn(M, K, device=dev).to(BF)
        W = (torch.randn(N, K, device=dev) * 0.02).to(BF)
        mb = N * K * 2 / 1e6
        if knobs: lib().eagle_debug_set_rows(*knobs)
        t_new = timed(lambda: ops.linear_rows(x, W))
        lib().eagle_debug_set_rows(0, 0, 0)
        t_old = timed(lambda: (ops.linear_skinny(x, W) if M <= 64 and N < 8192 else ops.gemm_nt(x, W)))
        print(f"M={M:4d} {name:8s} {mb:7.1f} MB  rows {t_new:7.1f} us {mb / t_new:5.2f} TB/s   old {t_old:7.1f} us {mb / t_old:5.2f} TB/s", flush=True)
    I, K = 12288, 4096
    x = torch.randn(M, K, device=dev).to(BF)
    Wgu = (torch.randn(2 * I, K, device=dev) * 0.02).to(BF)
    mb = 2 * I * K * 2 / 1e6
    if knobs: lib().eagle_debug_set_rows(*knobs)
    t_new = timed(lambda: ops.linear_rows(x, Wgu[:I], W2=Wgu[I:]))
    lib().eagle_debug_set_rows(0, 0, 0)
    t_old = timed(lambda: (ops.linear_skinny(x, Wgu[:I], W2=Wgu[I:]) if M <= 32 else ops.silu_mul(ops.gemm_nt(x, Wgu))))
    print(f"M={M:4d} {'gate|up':8s} {mb:7.1f} MB  rows {t_new:7.1f} us {mb / t_new:5.2f} TB/s   old {t_old:7.1f} us {mb / t_old:5.2f} TB/s", flush=True)

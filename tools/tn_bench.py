"""Weight-gradient GEMM dW = dY^T . X at the bench's shapes: TN kernel vs two transposes + NT kernel.  GPU box only."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, M, N in (("wo", 4096, 4096), ("wqkv", 12288, 4096), ("fc", 4096, 8192), ("wgu", 24576, 4096), ("wdown", 4096, 12288)):
    dY = torch.randn(K, M, device=dev).to(torch.bfloat16); X = torch.randn(K, N, device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    def t_tn(): ops.gemm_tn(dY, X, out=C, accumulate=True)
    def t_nt(): ops.gemm_nt(ops.transpose(dY, pad_to=64), ops.transpose(X, pad_to=64), out=C, accumulate=True)
    for fn, lab in ((t_tn, "TN"), (t_nt, "transposes + NT")):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"{name:6s} M={M} N={N} K={K} {lab:16s} {us:8.1f} us {2*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)

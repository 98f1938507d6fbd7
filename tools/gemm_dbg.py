"""Timing experiments on the 256x256 ping-pong GEMM (results invalid in DBG modes): which of DMA,
LDS reads and barriers holds the main loop back.  Run on the GPU box."""
import sys, ctypes, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eagle_amd import ops, _hip
dev = torch.device('cuda:0')
setm = _hip.lib().eagle_debug_set_gemm8; setm.argtypes, setm.restype = [ctypes.c_int], None
setpp = _hip.lib().eagle_debug_set_gemm8_pingpong; setpp.argtypes, setpp.restype = [ctypes.c_int], None
setm(1)
for M, N, K, name in [(4096, 151936, 4096, 'lm_head qwen'), (4096, 24576, 4096, 'gate|up'), (4096, 4096, 12288, 'down'), (8192, 8192, 8192, '8k cube')]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for mode, label in ((1, 'full'), (17, 'no DMA'), (18, 'no LDS reads'), (19, 'no DMA, no reads'), (23, 'MFMA only (no barriers)'), (24, 'DMA from 2 hot K tiles'), (32, 'DMA, no vmcnt wait'), (40, 'hot DMA, no wait'), (80, 'no output stores')):
        setpp(mode)
        for _ in range(3): ops.gemm_nt(A, B, out=C)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm_nt(A, B, out=C)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        extra = ""
        if mode == 1:
            ref = A[:512].float() @ B[:1024].float().t()
            extra = f"  max rel err {float((C[:512, :1024].float() - ref).abs().max() / ref.abs().max()):.2e}"
        print(f"{name:14s} {label:34s} {ms*1e3:9.1f} us {2*M*N*K/ms/1e9:8.1f} TFLOP/s{extra}", flush=True)

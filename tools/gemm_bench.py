"""Micro-benchmark of the training-size GEMMs (TFLOP/s), run on the GPU box."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device('cuda:0')
MT = int(sys.argv[1]) if len(sys.argv) > 1 else 2048          # tokens per micro-batch
shapes = [(2048, 4096, 8192, 'fc'), (2048, 12288, 4096, 'qkv'), (2048, 4096, 4096, 'o'), (2048, 24576, 4096, 'gate|up qwen'),
          (2048, 4096, 12288, 'down qwen'), (2048, 151936, 4096, 'lm_head qwen'), (2048, 4096, 151936, 'lm dgrad'),
          (4096, 12288, 2048, 'wgrad-like'), (8192, 4096, 4096, 'o 8k tokens')]
import ctypes
from eagle_amd import _hip
setm = _hip.lib().eagle_debug_set_gemm8
shapes = [(MT if M == 2048 else M, N, MT if K == 2048 else K, n) for M, N, K, n in shapes[:8]]
shapes += [(4096, 4096, MT, 'wgrad o'), (12288, 4096, MT, 'wgrad qkv'), (4096, 8192, MT, 'wgrad fc'), (24576, 4096, MT, 'wgrad gate|up'), (4096, 12288, MT, 'wgrad down')]
for M, N, K, name in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    it = 10
    for mode, pp in ((0, 0), (1, 1)):
        setm(mode)
        for _ in range(3): ops.gemm_nt(A, B, out=C)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): ops.gemm_nt(A, B, out=C)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        print(f"{name:16s} M={M} N={N} K={K} [{('256x256 ping-pong' if pp else '256x256 in-phase ') if mode else '128x128          '}]: {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
    setm(2)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); 
    for _ in range(it): torch.matmul(A, B.t(), out=C)
    t1.record(); torch.cuda.synchronize()
    print(f"{'  (hipBLASLt via torch, for reference)':16s} {t0.elapsed_time(t1)/it*1e3:9.1f} us  {2*M*N*K/(t0.elapsed_time(t1)/it)/1e9:8.1f} TFLOP/s", flush=True)

"""The training step's dominant GEMMs at 4096 tokens: ours vs hipBLASLt (torch.matmul), interleaved rounds, random operands."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eagle_amd import ops
dev = torch.device('cuda:0')
shapes = [(4096, 151936, 4096, 'lm_head fwd'), (4096, 4096, 151936, 'lm_head dgrad'), (4096, 24576, 4096, 'gate|up'), (4096, 12288, 4096, 'qkv')]
for M, N, K, name in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ks = 1
    if name == 'lm_head dgrad':
        from eagle_amd.train import lm_dgrad_ksplit
        ks = lm_dgrad_ksplit(M, N, K)
    from eagle_amd._hip import lib
    gms = [int(x, 0) for x in sys.argv[1:]] or [0x20000, 0x20001, 0x20002]      # eagle_debug_set_gemm8 words: 0x20000 | schedule, 0x30000 | nt stores (1 on, 2 off)
    res = {f"ours knob={g:#x}": [] for g in gms}
    res["hipblaslt"] = []
    for rnd in range(4):
        for which in res:
            if which != "hipblaslt":
                lib().eagle_debug_set_gemm8(int(which.split("=")[1], 0))
            fn = (lambda: ops.gemm_nt(A, B, out=C, ksplit=ks)) if which != "hipblaslt" else (lambda: torch.matmul(A, B.t(), out=C))
            for _ in range(2): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            res[which].append(e0.elapsed_time(e1) / 5)
    for which, v in res.items():
        v.sort()
        print(f"{name:14s} {M}x{N}x{K} {which:18s}: median {v[len(v)//2]*1e3:8.1f} us = {2*M*N*K/v[len(v)//2]/1e9:7.1f} TFLOP/s   (min {v[0]*1e3:8.1f} us)", flush=True)
lib().eagle_debug_set_gemm8(0x20000 | 2)
lib().eagle_debug_set_gemm8(0x30000)

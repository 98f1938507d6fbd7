"""Tile-order group size (row tiles swept before moving one column tile on) of the 256x256 GEMM (GPU box)."""
import sys, ctypes, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eagle_amd import ops, _hip
dev = torch.device('cuda:0')
setm = _hip.lib().eagle_debug_set_gemm8; setm.argtypes, setm.restype = [ctypes.c_int], None
setgm = _hip.lib().eagle_debug_set_gemm8_gm; setgm.argtypes, setgm.restype = [ctypes.c_int], None
setm(1)
MT = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for M, N, K, name in [(MT, 151936, 4096, 'lm_head'), (MT, 24576, 4096, 'gate|up'), (MT, 12288, 4096, 'qkv'), (MT, 4096, 12288, 'down'),
                      (MT, 4096, 151936, 'lm dgrad'), (24576, 4096, MT, 'wgrad gate|up'), (4096, 12288, MT, 'wgrad down')]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = A[:300].float() @ B[:700].float().t()
    line = f"{name:14s} M={M} N={N} K={K}:"
    for gm in (0, 2, 4, 8):
        setgm(gm)
        for _ in range(3): ops.gemm_nt(A, B, out=C)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm_nt(A, B, out=C)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        err = float((C[:300, :700].float() - ref).abs().max() / ref.abs().max())
        line += f"  gm={gm}: {2*M*N*K/ms/1e9:7.1f} TF ({err:.0e})"
    print(line, flush=True)

"""Which kernel for a decode-batch projection of M rows (GPU box): weight-stream (skinny) vs the MFMA GEMMs."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops, _hip
from eagle_amd.llama2 import _gemm_wide
dev = torch.device("cuda:0")
setm = _hip.lib().eagle_debug_set_gemm8
scrub = torch.empty(1 << 27, dtype=torch.float32, device=dev)

def timeit(fn, it=6):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for i in range(it):
        scrub.fill_(float(i))                      # 512 MiB write: the weights do not stay in the Infinity Cache
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / it * 1e3

shapes = [(151936, 4096, "lm_head"), (4096, 8192, "fc"), (12288, 4096, "qkv"), (4096, 4096, "o"), (24576, 4096, "gate|up"), (4096, 12288, "down")]
for N, K, name in shapes:
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    for M in (32, 64, 128, 192):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        r = {}
        r["skinny"] = timeit(lambda: ops.linear_skinny(x, W))
        setm(2); r["wide(auto)"] = timeit(lambda: _gemm_wide(x, W))
        setm(1); r["nt8 k1"] = timeit(lambda: ops.gemm_nt(x, W))
        setm(0); r["nt128"] = timeit(lambda: ops.gemm_nt(x, W))
        setm(2)
        gb = N * K * 2 / 1e9
        print(f"{name:8s} M={M:4d} " + "  ".join(f"{k}: {v:7.1f} us ({gb / v * 1e3:4.2f} TB/s)" for k, v in r.items()), flush=True)
    del W

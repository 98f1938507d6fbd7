"""PMC target: the row-batch weight stream (gemm_rows_stream_kernel) on the Qwen3 lm_head (151936 x 4096 bf16) at 32 and at
128 rows, from cold caches, and nothing else of note -- FETCH_SIZE per launch tells whether the x re-reads of the column
tiles reach HBM (they should not: x is 256 KB / 1 MB and L2-resident).
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_rows -- python3 tools/prof_rows.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
V, D = 151936, 4096
g = torch.Generator(device=dev).manual_seed(0)
W = (torch.randn((V, D), generator=g, device=dev) * 0.02).to(torch.bfloat16)
scrub = torch.empty(1 << 28, dtype=torch.float32, device=dev)
for M in (32, 128):
    x = torch.randn((M, D), generator=g, device=dev).to(torch.bfloat16)
    out = torch.empty((M, V), device=dev, dtype=torch.bfloat16)
    for i in range(4):
        scrub.fill_(float(i))
        ops.linear_rows(x, W, out=out)
torch.cuda.synchronize()
print("done")

"""PMC target: N launches of the dominant decode kernel (the lm_head weight stream of bench.py's
`roofline`: gemm_skinny_keepx_kernel<1,0,4,...,PK,TKP>, 4 rows x 32000 x 4096 bf16 over the packed image, top-k candidate epilogue;
--logits: the same stream writing bf16 logits, --row-major: over the row-major weight) and nothing else of note, so that
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE rows of that kernel can be read per launch.
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/prof_lm_head.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops  # noqa: E402

V, D, rows = (151936, 4096, 4) if "--qwen" in sys.argv else (32000, 4096, 4)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
W = (torch.randn((V, D), generator=g, device=dev) * 0.02).to(torch.bfloat16)
x = torch.randn((rows, D), generator=g, device=dev).to(torch.bfloat16)
out = torch.empty((rows, V), device=dev, dtype=torch.bfloat16)
# a 1 GiB scrub between launches keeps the 256 MiB Infinity Cache from serving the re-read
scrub = torch.empty(1 << 28, dtype=torch.float32, device=dev)
Wp = ops.pack_weight(W)          # what the drafter streams: the packed image (whole-cache-line reads)
ti = torch.zeros((1, rows, 10), dtype=torch.int64, device=dev)
tv = torch.zeros((1, rows, 10), dtype=torch.bfloat16, device=dev)
for i in range(6):
    scrub.fill_(float(i))
    if "--row-major" in sys.argv or "--logits" in sys.argv:
        ops.linear_skinny(x, Wp if "--row-major" not in sys.argv else W, out=out)
    else:
        ops.lm_head_topk_into_table(x, Wp, 10, ti, tv, 0, rows)      # what the one-request drafter launches per tree step
torch.cuda.synchronize()
print("done", float(out.float().abs().mean()))

"""PMC target: the training lm_head GEMM (4096 x 151936 x 4096 bf16, gemm_nt8_kernel ping-pong) a few times.
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d ... -- python3 tools/prof_gemm.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = 4096, 151936, 4096
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
B = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    ops.gemm_nt(A, B, out=C)
torch.cuda.synchronize()
print("done", float(C[:4, :64].float().abs().mean()))

"""Calibration only: which library kernels torch.matmul dispatches for the 7B shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
M = 22528
for m, n, k in [(M, 12288, 4096), (M, 4096, 4096), (M, 22016, 4096), (M, 4096, 11008)]:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()

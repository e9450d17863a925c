"""Micro-benchmark of rv_gemm_bf16 operand forms on training shapes (random data)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops
lib.load().rv_gemm_select_kernel(2)
T = 22528
CASES = [("NT", 8192, 4096, 4096, False, False), ("NN dgrad", 8192, 4096, 4096, False, True), ("TN", 8192, 4096, 4096, True, False),
         ("TT wgrad", 4096, 4096, 8192, True, True),
         ("gu_dgrad NN", T, 4096, 22016, False, True), ("gu_wgrad TT", 22016, 4096, T, True, True), ("o_wgrad TT", 4096, 4096, T, True, True)]
for name, m, n, k, ta, tb in CASES:
    a = torch.randn((k, m) if ta else (m, k), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((k, n) if tb else (n, k), device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, b, ta=ta, tb=tb, out=c)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm(a, b, ta=ta, tb=tb, out=c)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print(name, (m, n, k), round(2.0 * m * n * k / (best * 1e-3) / 1e12, 1), "TF/s", flush=True)

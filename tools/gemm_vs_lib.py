"""Calibration only (not product): this repo's GEMM vs the vendor library (torch.matmul -> hipBLASLt/rocBLAS) on the 7B shapes,
same process, interleaved, random N(0,1) data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops

M = 22528
SHAPES = [("qkv_fwd", M, 12288, 4096), ("o_fwd", M, 4096, 4096), ("gu_fwd", M, 22016, 4096), ("down_fwd", M, 4096, 11008),
          ("lm_head", M, 32000, 4096)]
lib.load()


def t(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, m, n, k in SHAPES:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    best = [1e9, 1e9]
    for rnd in range(3):
        best[0] = min(best[0], t(lambda: ops.gemm(a, b, out=c)))
        best[1] = min(best[1], t(lambda: torch.matmul(a, b.t(), out=c)))
    f = 2.0 * m * n * k / 1e9
    print(f"{name} {(m, n, k)}: ours {f / best[0]:.0f} TF/s, library {f / best[1]:.0f} TF/s", flush=True)

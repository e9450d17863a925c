"""A/B micro-benchmark of the GEMM kernels on the 7B training shapes (one process, interleaved rounds, random data)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops

M = 22528
SHAPES = [("qkv_fwd", M, 12288, 4096), ("o_fwd", M, 4096, 4096), ("gu_fwd", M, 22016, 4096), ("down_fwd", M, 4096, 11008),
          ("qkv_wgrad", 12288, 4096, M), ("gu_wgrad", 22016, 4096, M), ("down_wgrad", 4096, 11008, M), ("o_wgrad", 4096, 4096, M),
          ("lm_head", M, 32000, 4096), ("gu_dgrad", M, 4096, 22016)]
kernels = [int(k) for k in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2"])]
l = lib.load()
res = {}
for name, m, n, k in SHAPES:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    best = {kk: 1e9 for kk in kernels}
    for rnd in range(3):
        for kk in kernels:
            l.rv_gemm_select_kernel(kk)
            ops.gemm_nt(a, b, out=c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_nt(a, b, out=c)
            e1.record()
            torch.cuda.synchronize()
            best[kk] = min(best[kk], e0.elapsed_time(e1) / 5)
    res[name] = {kk: round(2.0 * m * n * k / (best[kk] * 1e-3) / 1e12, 1) for kk in kernels}
    print(name, (m, n, k), res[name], flush=True)
    del a, b, c
l.rv_gemm_select_kernel(0)

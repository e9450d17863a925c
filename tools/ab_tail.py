"""Same-box A/B of the tail-round K-split (rv_gemm_select_kernel 20 = off, 21 = on)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops
l = lib.load()
T = 22528
CASES = [("qkv_fwd", T, 12288, 4096, 0, 0), ("o_fwd", T, 4096, 4096, 0, 0), ("down_fwd", T, 4096, 11008, 0, 0),
         ("dh1 NN", T, 4096, 12288, 0, 1), ("dattn NN", T, 4096, 4096, 0, 1), ("dh2 NN", T, 4096, 22016, 0, 1),
         ("gu_wgrad TT", 22016, 4096, T, 1, 1), ("gu_fwd (n/a)", T, 22016, 4096, 0, 0)]
for name, m, n, k, ta, tb in CASES:
    a = torch.randn((k, m) if ta else (m, k), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((k, n) if tb else (n, k), device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    best = {20: 1e9, 21: 1e9}
    for rnd in range(4):
        for kk in (20, 21):
            l.rv_gemm_select_kernel(kk)
            ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=c)
            e1.record(); torch.cuda.synchronize()
            best[kk] = min(best[kk], e0.elapsed_time(e1) / 5)
    fl = 2.0 * m * n * k
    print(f"{name:14s} off {fl/best[20]/1e9:7.1f}  on {fl/best[21]/1e9:7.1f} TF/s  speedup {best[20]/best[21]:.3f}", flush=True)
l.rv_gemm_select_kernel(21)

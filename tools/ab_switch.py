"""Same-binary A/B of a rv_gemm_select_kernel switch on the bench's GEMM shapes (interleaved rounds, random data, best of 4 x 5 launches).
    python tools/ab_switch.py 30 31      -> A = select(30) [buffer-addressed staging off], B = select(31) [on]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops

sa, sb = int(sys.argv[1]), int(sys.argv[2])
l = lib.load()
T = 22528
CASES = [("sq4096", 4096, 4096, 4096, 0, 0), ("sq8192", 8192, 8192, 8192, 0, 0), ("qkv_fwd", T, 12288, 4096, 0, 0), ("o_fwd", T, 4096, 4096, 0, 0),
         ("gu_fwd", T, 22016, 4096, 0, 0), ("down_fwd", T, 4096, 11008, 0, 0), ("head_fwd", T, 32000, 4096, 0, 0),
         ("dh2 NN", T, 4096, 22016, 0, 1), ("dact NN", T, 11008, 4096, 0, 1), ("dx_o NN", T, 4096, 4096, 0, 1),
         ("gu_wgrad TT", 22016, 4096, T, 1, 1), ("down_wgrad TT", 4096, 11008, T, 1, 1), ("o_wgrad TT", 4096, 4096, T, 1, 1), ("qkv_wgrad TT", 12288, 4096, T, 1, 1)]
tot = {sa: 0.0, sb: 0.0}
for name, m, n, k, ta, tb in CASES:
    a = torch.randn((k, m) if ta else (m, k), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((k, n) if tb else (n, k), device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    run = lambda: ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=c)
    l.rv_gemm_select_kernel(sa); run(); ref = c.clone()
    l.rv_gemm_select_kernel(sb)
    for _ in range(3):
        c.zero_(); run()
        if sa >= 30:
            assert torch.equal(c, ref), f"{name}: B differs from A"
        else:       # launch-shape switches change the fp32 summation order: equal up to bf16 rounding of a few elements
            assert float((c.float() - ref.float()).abs().max()) <= 2.0 ** -6 * float(ref.float().abs().max()), name
    best = {sa: 1e9, sb: 1e9}
    for rnd in range(4):
        for key in (sa, sb):
            l.rv_gemm_select_kernel(key)
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            best[key] = min(best[key], e0.elapsed_time(e1) / 5)
    fl = 2.0 * m * n * k
    tot[sa] += best[sa]; tot[sb] += best[sb]
    print(f"{name:14s} A {fl/best[sa]/1e9:7.1f} TF/s   B {fl/best[sb]/1e9:7.1f} TF/s   B/A {best[sa]/best[sb]:.3f}", flush=True)
print(f"sum of times: A {tot[sa]:.2f} ms  B {tot[sb]:.2f} ms  B/A speed {tot[sa]/tot[sb]:.3f}")
l.rv_gemm_select_kernel(31)

"""Fused-epilogue GEMMs at the bench shapes: time + output checksum of one library build (RADVLM_HIP_LIB selects it); run once per build
and compare (tools/ab_fused.sh).  python tools/ab_fused.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops
lib.load()
M, d, F = 22528, 4096, 11008
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, std=1.0: (torch.randn(*s, generator=g, device="cuda") * std).to(torch.bfloat16)
dy, wd, gu, x, wgu = rn(M, d, std=0.01), rn(d, F, std=0.02), rn(M, 2 * F), rn(M, d), rn(2 * F, d, std=0.02)
def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
dgu = torch.empty_like(gu)
t = timeit(lambda: ops.gemm_swiglu_bwd(dy, wd, gu, F, dgu=dgu))
print(f"swiglu_bwd  {t*1e3:8.1f} us  {2.0*M*F*d/t/1e9:7.1f} TF/s  checksum {float(dgu.double().sum()):.10e} {float(dgu.double().abs().sum()):.10e}")
out = ops.gemm_swiglu_fwd(x, wgu, F)
t = timeit(lambda: ops.gemm_swiglu_fwd(x, wgu, F))
print(f"swiglu_fwd  {t*1e3:8.1f} us  {2.0*M*2*F*d/t/1e9:7.1f} TF/s  checksum {float(out[1].double().sum()):.10e} {float(out[0].double().abs().sum()):.10e}")

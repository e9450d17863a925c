import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops
lib.load().rv_gemm_select_kernel(2)
M, N = 22528, 22016
for K in (64, 128, 256, 1024, 4096):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt(a, b, out=c); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm_nt(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print(f"K={K}: {t*1e3:.0f} us  ({2.0*M*N*K/t/1e9:.0f} TF/s); C write {M*N*2/1e6:.0f} MB -> {M*N*2/t/1e6:.0f} GB/s")
print("fp32 output (16-byte stores per lane):")
for K in (64, 4096):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm_nt(a, b, out=c); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm_nt(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print(f"K={K}: {t*1e3:.0f} us  ({2.0*M*N*K/t/1e9:.0f} TF/s); C write {M*N*4/1e6:.0f} MB -> {M*N*4/t/1e6:.0f} GB/s")

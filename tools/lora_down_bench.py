"""rv_lora_down_bf16 (fused dropout + r-wide GEMM) vs the two-launch sequence, 13B LoRA shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import ops


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = 22528
for K in (5120, 13824):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    a = torch.randn(64, K, device="cuda", dtype=torch.bfloat16) * 0.02
    for p in (0.05, 0.0):
        fused = t(lambda: ops.lora_down(x, a, 0.25, p, 7))
        two = t(lambda: ops.gemm(ops.dropout(x, p, 7) if p > 0 else x, a, alpha=0.25))
        gb = M * K * 2 / 1e9
        print(f"K={K} p={p}: fused {fused:7.1f} us ({gb / fused * 1e3:5.2f} TB/s of x)   dropout + gemm {two:7.1f} us", flush=True)

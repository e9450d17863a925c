"""Micro-benchmark of the attention kernels at the 7B training shape (B=32, H=32, S=704, hd=128, causal)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops
B, H, S, hd = 32, 32, 704, 128
d = H * hd
qkv = torch.randn(B * S, 3 * d, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(B * S, d, device="cuda", dtype=torch.bfloat16)
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
vT = ops.transpose_heads(v, B, S, H, hd, S)
out, lse = ops.attn_fwd(q, k, vT, B, S, H, hd, S, True, lens=lens)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_f = timeit(lambda: ops.attn_fwd(q, k, vT, B, S, H, hd, S, True, lens=lens, out=out, lse=lse))
dq, dk, dv = torch.empty_like(dout), torch.empty_like(dout), torch.empty_like(dout)
t_b = timeit(lambda: ops.attn_bwd(q, k, v, out, dout, lse, B, S, H, hd, S, True, lens=lens, dq=dq, dk=dk, dv=dv))
fl_f = 4.0 * B * H * S * S * hd / 2
print(f"attn fwd {t_f*1e3:.0f} us ({fl_f/t_f/1e9:.0f} TF/s causal-algorithmic), bwd (incl. transposes, delta) {t_b*1e3:.0f} us ({2.5*fl_f/t_b/1e9:.0f} TF/s)")

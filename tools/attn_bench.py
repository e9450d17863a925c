"""Micro-benchmark of the attention kernels (causal, hd=128): the 7B training shape and the long-sequence shapes of configs 4 / 8f.1.
    python tools/attn_bench.py            -> fwd / bwd time and causal-algorithmic TFLOP/s per shape (bwd includes its transposes / delta)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops

hd = 128


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, H, Hkv, S in ((32, 32, 32, 704), (8, 32, 32, 3056), (2, 28, 4, 7499)):
    d, kvd = H * hd, Hkv * hd
    s_pad = (S + 63) // 64 * 64
    qkv = torch.randn(B * S, d + 2 * kvd, device="cuda", dtype=torch.bfloat16)
    dout = torch.randn(B * S, d, device="cuda", dtype=torch.bfloat16)
    q, k, v = qkv[:, :d], qkv[:, d:d + kvd], qkv[:, d + kvd:]
    vT = ops.transpose_heads(v, B, S, Hkv, hd, s_pad)
    out, lse = ops.attn_fwd(q, k, vT, B, S, H, hd, s_pad, True, kv_heads=Hkv)
    t_f = timeit(lambda: ops.attn_fwd(q, k, vT, B, S, H, hd, s_pad, True, kv_heads=Hkv, out=out, lse=lse))
    t_n = timeit(lambda: ops.attn_fwd(q, k, None, B, S, H, hd, s_pad, True, kv_heads=Hkv, out=out, lse=lse, v=v))
    print(f"   natural-layout fwd {t_n*1e3:.0f} us ({4.0 * B * H * S * S * hd / 2/t_n/1e9:.0f} TF/s)")
    t_t = timeit(lambda: ops.transpose_heads(v, B, S, Hkv, hd, s_pad, out=vT))
    dq, dk, dv = torch.empty_like(dout), torch.empty(B * S, kvd, device="cuda", dtype=torch.bfloat16), torch.empty(B * S, kvd, device="cuda", dtype=torch.bfloat16)
    t_b = timeit(lambda: ops.attn_bwd(q, k, v, out, dout, lse, B, S, H, hd, s_pad, True, kv_heads=Hkv, dq=dq, dk=dk, dv=dv, natural=False))
    t_bn = timeit(lambda: ops.attn_bwd(q, k, v, out, dout, lse, B, S, H, hd, s_pad, True, kv_heads=Hkv, dq=dq, dk=dk, dv=dv, natural=True))
    print(f"   natural-layout bwd {t_bn*1e3:.0f} us ({2.5 * 4.0 * B * H * S * S * hd / 2/t_bn/1e9:.0f} TF/s = {2.5 * 4.0 * B * H * S * S * hd / 2/t_bn/1e9/25:.1f} %)")
    fl_f = 4.0 * B * H * S * S * hd / 2
    print(f"B={B} H={H}:{Hkv} S={S}: fwd {t_f*1e3:.0f} us ({fl_f/t_f/1e9:.0f} TF/s = {fl_f/t_f/1e9/25:.1f} % of peak) + V^T copy {t_t*1e3:.0f} us; "
          f"bwd incl. transposes {t_b*1e3:.0f} us ({2.5*fl_f/t_b/1e9:.0f} TF/s = {2.5*fl_f/t_b/1e9/25:.1f} %)", flush=True)

"""Does the row-major [token, head * 128] layout (a head's 256-byte slice of a token row every 24 KB) cost the attention kernels HBM
efficiency?  Times the SAME kernels on the same amount of data laid out head-major (each (sample, head) a contiguous [S, 128] block), which the
entry points take as B * H samples of one head with a row stride of 128 elements.    python tools/attn_layout_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import ops

hd = 128


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, H, S in ((32, 32, 704), (8, 32, 3056)):
    s_pad = (S + 63) // 64 * 64
    res = {}
    cases = [("token-major [rows, 3 * H * 128]", (B, H), 0), ("token-major, q / k / v apart    ", (B, H), -1), ("head-major  [(b, h), S, 128]   ", (B * H, 1), -1)]
    cases += [(f"token-major, row stride + {pad:4d} el", (B, H), pad) for pad in (64, 128, 256, 512, 1024, 2048)]
    for name, (b_, h_), pad in cases:
        d = h_ * hd
        if pad >= 0:
            qkv = torch.randn(b_ * S, 3 * d + pad, device="cuda", dtype=torch.bfloat16)
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:3 * d]
        else:
            q, k, v = (torch.randn(b_ * S, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        dout = torch.randn(b_ * S, d, device="cuda", dtype=torch.bfloat16)
        out, lse = ops.attn_fwd(q, k, None, b_, S, h_, hd, s_pad, True, kv_heads=h_, v=v)
        t_f = timeit(lambda: ops.attn_fwd(q, k, None, b_, S, h_, hd, s_pad, True, kv_heads=h_, out=out, lse=lse, v=v))
        dq, dk, dv = torch.empty_like(dout), torch.empty_like(dout), torch.empty_like(dout)
        t_b = timeit(lambda: ops.attn_bwd(q, k, v, out, dout, lse, b_, S, h_, hd, s_pad, True, kv_heads=h_, dq=dq, dk=dk, dv=dv, natural=True))
        print(f"B={B} H={H} S={S}  {name}: fwd {t_f:.0f} us, bwd {t_b:.0f} us", flush=True)

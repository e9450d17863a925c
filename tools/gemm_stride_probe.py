"""Do the activations' row strides (4096 / 11008 / 12288 / 22016 elements: multiples of 8 KB in three of four cases) cost the GEMMs anything?  The decoder's
shapes with every ACTIVATION operand / output stored with its natural row stride and with the stride padded by 64 elements (128 bytes); weights stay
contiguous (they are views of the flat parameter buffer).  Same box, interleaved, best of 3 rounds x 5 launches.   python tools/gemm_stride_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import ops

T, D, F = 22528, 4096, 11008
PAD = 64


def t(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def act(rows, cols, pad):
    return torch.randn(rows, cols + pad, device="cuda", dtype=torch.bfloat16)[:, :cols]


CASES = [("qkv fwd", T, 3 * D, D, 0, 0), ("o fwd", T, D, D, 0, 0), ("gate|up fwd", T, 2 * F, D, 0, 0), ("down fwd", T, D, F, 0, 0),
         ("qkv dgrad", T, D, 3 * D, 0, 1), ("o dgrad", T, D, D, 0, 1), ("gate|up dgrad", T, D, 2 * F, 0, 1), ("down dgrad", T, F, D, 0, 1),
         ("qkv wgrad", 3 * D, D, T, 1, 1), ("o wgrad", D, D, T, 1, 1), ("gate|up wgrad", 2 * F, D, T, 1, 1), ("down wgrad", D, F, T, 1, 1)]
tot = [0.0, 0.0]
for name, m, n, k, ta, tb in CASES:
    res = []
    for pad in (0, PAD):
        if ta:      # weight gradient: both operands are activations stored [tokens, features]; the output is a (contiguous) gradient view
            a, b, c = act(k, m, pad), act(k, n, pad), torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        elif tb:    # input gradient: dy [tokens, out] activation, W [out, in] contiguous weight, dx activation
            a, b, c = act(m, k, pad), torch.randn(k, n, device="cuda", dtype=torch.bfloat16) * 0.02, act(m, n, pad)
        else:       # forward: x activation, W [out, in] weight, y activation
            a, b, c = act(m, k, pad), torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.02, act(m, n, pad)
        res.append((a, b, c))
    best = [1e9, 1e9]
    for _ in range(3):
        for i, (a, b, c) in enumerate(res):
            best[i] = min(best[i], t(lambda: ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=c)))
    fl = 2.0 * m * n * k
    tot[0] += best[0]; tot[1] += best[1]
    print(f"{name:14s} ({m}, {n}, {k}): natural strides {best[0]:7.0f} us ({fl / best[0] / 1e6:5.0f} TF/s)   + {PAD} elements {best[1]:7.0f} us ({fl / best[1] / 1e6:5.0f} TF/s)   "
          f"ratio {best[0] / best[1]:.3f}", flush=True)
print(f"sum: natural {tot[0] / 1e3:.2f} ms, padded {tot[1] / 1e3:.2f} ms, ratio {tot[0] / tot[1]:.3f}")

"""Same-box table: this repo's GEMM (`ops.gemm`, product launch path: persistent blocks, tail split) vs the vendor library
(`torch.matmul` -> hipBLASLt / rocBLAS) on the 14 GEMM shapes of one LLaVA-1.5-7B decoder layer + head at the bench's M = 22528 token rows,
forward (row x row), input-gradient (row x contraction-major) and weight-gradient (both contraction-major) forms, interleaved rounds, random
N(0,1) data, best of 4 rounds x 5 launches.  Calibration only (the product never calls the vendor library).

    python tools/gemm_decoder_shapes.py [out.json]
    python tools/gemm_decoder_shapes.py --recipe [out.json]     the RadVLM recipe's shapes instead (Qwen2-7B widths, M = 14998 token rows of
                                                                the anyres_max_9 batch: not a multiple of the 256-row tile), with the tile
                                                                accounting per shape: edge-tile fill and how full the last round of 256 CUs is"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops

T = 22528
CASES = [("qkv fwd", T, 12288, 4096, 0, 0), ("o fwd", T, 4096, 4096, 0, 0), ("gate|up fwd", T, 22016, 4096, 0, 0), ("down fwd", T, 4096, 11008, 0, 0),
         ("lm_head fwd", T, 32000, 4096, 0, 0),
         ("qkv dgrad", T, 4096, 12288, 0, 1), ("o dgrad", T, 4096, 4096, 0, 1), ("gate|up dgrad", T, 4096, 22016, 0, 1), ("down dgrad", T, 11008, 4096, 0, 1),
         ("lm_head dgrad", T, 4096, 32000, 0, 1),
         ("qkv wgrad", 12288, 4096, T, 1, 1), ("o wgrad", 4096, 4096, T, 1, 1), ("gate|up wgrad", 22016, 4096, T, 1, 1), ("down wgrad", 4096, 11008, T, 1, 1)]
RECIPE = "--recipe" in sys.argv
if RECIPE:
    sys.argv.remove("--recipe")
    T, D, KV, FF, VOC = 14998, 3584, 512, 18944, 152064
    CASES = [("qkv fwd", T, D + 2 * KV, D, 0, 0), ("o fwd", T, D, D, 0, 0), ("gate|up fwd", T, 2 * FF, D, 0, 0), ("down fwd", T, D, FF, 0, 0),
             ("lm_head fwd", T, VOC, D, 0, 0),
             ("qkv dgrad", T, D, D + 2 * KV, 0, 1), ("o dgrad", T, D, D, 0, 1), ("gate|up dgrad", T, D, 2 * FF, 0, 1), ("down dgrad", T, FF, D, 0, 1),
             ("lm_head dgrad", T, D, VOC, 0, 1),
             ("qkv wgrad", D + 2 * KV, D, T, 1, 1), ("o wgrad", D, D, T, 1, 1), ("gate|up wgrad", 2 * FF, D, T, 1, 1), ("down wgrad", D, FF, T, 1, 1),
             ("lm_head wgrad", VOC, D, T, 1, 1)]
lib.load()


def tiles(m, n, cus=256):
    """256 x 256 output tiles: count, useful fraction of the edge-padded tile area, rounds over the CUs and the fill of the last round."""
    tm, tn = -(-m // 256), -(-n // 256)
    nt = tm * tn
    return dict(tiles=nt, useful_area=m * n / (tm * tn * 65536.0), rounds=nt / cus, last_round_fill=(nt % cus) / cus if nt % cus else 1.0)


def t(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows, tot = [], [0.0, 0.0]
for name, m, n, k, ta, tb in CASES:
    a = torch.randn((k, m) if ta else (m, k), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((k, n) if tb else (n, k), device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    av, bv = (a.t() if ta else a), (b if tb else b.t())
    best = [1e9, 1e9]
    for rnd in range(4):
        best[0] = min(best[0], t(lambda: ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=c)))
        best[1] = min(best[1], t(lambda: torch.matmul(av, bv, out=c)))
    f = 2.0 * m * n * k / 1e9
    tot[0] += best[0]
    tot[1] += best[1]
    rows.append(dict(shape=name, m=m, n=n, k=k, form="TT" if ta and tb else ("NN" if tb else "NT"), ours_tflops=f / best[0], library_tflops=f / best[1],
                     ours_us=best[0] * 1e3, library_us=best[1] * 1e3, **tiles(m, n)))
    tl = tiles(m, n)
    print(f"{name:16s} {(m, n, k)}: ours {f / best[0]:7.0f} TF/s   library {f / best[1]:7.0f} TF/s   ours/library {best[1] / best[0]:.3f}   "
          f"tiles {tl['tiles']:5d} = {tl['rounds']:.2f} rounds (last {100 * tl['last_round_fill']:.0f} % full), useful tile area {100 * tl['useful_area']:.1f} %", flush=True)
out = dict(device=torch.cuda.get_device_name(0), token_rows=T, rows=rows, sum_ms_ours=tot[0], sum_ms_library=tot[1], ours_over_library_speed=tot[1] / tot[0])
print(f"sum over the {len(CASES)} shapes: ours {tot[0]:.2f} ms, library {tot[1]:.2f} ms, speed ratio {tot[1] / tot[0]:.3f}")
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

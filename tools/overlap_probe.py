"""Does an HBM-bound kernel (AdamW) on a side stream make progress beside the MFMA-bound persistent GEMMs of the main stream?
Serial vs overlapped wall time, for several GEMM CU budgets (rv_gemm_set_cu_budget reserves units for the side stream).
    python tools/overlap_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops

L = lib.load()
dev = "cuda:0"
n = 1 << 30                      # 1.07e9 parameters: 30 GB of AdamW traffic (~5 ms)
p = torch.zeros(n, dtype=torch.bfloat16, device=dev)
g = torch.full((n,), 1e-3, dtype=torch.bfloat16, device=dev)
master, m, v = (torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(3))
T = 22528
a = torch.randn(T, 4096, device=dev, dtype=torch.bfloat16)
w = torch.randn(12288, 4096, device=dev, dtype=torch.bfloat16)
c = torch.empty(T, 12288, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream()
NG = 12


def gemms():
    for _ in range(NG):
        ops.gemm(a, w, out=c)


def adam():
    ops.adamw(p, master, g, m, v, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1)


def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def overlapped():
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        adam()
    gemms()
    torch.cuda.current_stream().wait_stream(side)


for reserved in (0, 8, 16, 32):
    L.rv_gemm_set_cu_budget(0, reserved)
    for persist in (41, 40):
        L.rv_gemm_select_kernel(persist)
        tg, ta = wall(gemms), wall(adam)
        to = wall(overlapped)
        print(f"reserved CUs {reserved:3d} persistent={persist == 41}: {NG} GEMMs {tg:7.2f} ms, AdamW {ta:6.2f} ms, serial {tg + ta:7.2f} ms, overlapped {to:7.2f} ms "
              f"(hidden {100 * (tg + ta - to) / ta:5.1f} % of AdamW)", flush=True)
L.rv_gemm_set_cu_budget(0, 0)
L.rv_gemm_select_kernel(41)

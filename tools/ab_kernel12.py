"""Same-box comparison of the 128x128 (1) and 256x256 (2) kernels on short-K / awkward shapes (NT form)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops
l = lib.load()
CASES = [("vit_out", 18464, 1024, 1024), ("vit_fc2", 18464, 1024, 4096), ("vit_fc1", 18464, 4096, 1024), ("vit_qkv", 18464, 3072, 1024),
         ("proj0", 18432, 4096, 1024), ("proj2", 18432, 4096, 4096), ("patch", 18432, 1024, 592), ("o_fwd", 22528, 4096, 4096), ("b8 qkv", 5632, 12288, 4096),
         ("b8 o", 5632, 4096, 4096)]
for name, m, n, k in CASES:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16); b = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    best = {1: 1e9, 2: 1e9}
    for rnd in range(3):
        for kk in (1, 2):
            l.rv_gemm_select_kernel(kk)
            ops.gemm_nt(a, b, out=c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gemm_nt(a, b, out=c)
            e1.record(); torch.cuda.synchronize()
            best[kk] = min(best[kk], e0.elapsed_time(e1) / 5)
    fl = 2.0 * m * n * k
    t256 = ((m + 255) // 256) * ((n + 255) // 256)
    print(f"{name:8s} ({m},{n},{k}) tiles256={t256:5d}  k1 {fl/best[1]/1e9:7.1f}  k2 {fl/best[2]/1e9:7.1f} TF/s", flush=True)
l.rv_gemm_select_kernel(0)

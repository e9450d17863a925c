"""rmsnorm fwd/bwd bandwidth at the 7B shape (22528 x 4096)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import ops
rows, d = 22528, 4096
x = torch.randn(rows, d, device="cuda", dtype=torch.bfloat16); dy = torch.randn_like(x); dx = torch.randn_like(x)
w = torch.ones(d, device="cuda", dtype=torch.bfloat16); dw = torch.zeros_like(w)
y, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
b = rows * d * 2
tf = t(lambda: ops.rmsnorm_fwd(x, w, 1e-5, y=y, rstd=rstd)); tb = t(lambda: ops.rmsnorm_bwd(dy, x, w, rstd, dx=dx, dx_add=True, dw=dw))
print(f"rmsnorm fwd {tf:.0f} us ({2 * b / tf / 1e6:.2f} TB/s), bwd+dw (dx accumulate) {tb:.0f} us ({4 * b / tb / 1e6:.2f} TB/s)")

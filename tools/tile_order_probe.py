"""Three launches each of three bench GEMM shapes (NT / NN / TT) -- run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` with
RADVLM_HIP_LIB pointing at a build variant (-DRV_GROUP_M=...) to compare the block -> tile orders by L2-miss traffic and time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops

lib.load().rv_gemm_select_kernel(20)          # plain launch shape (no tail split): one kernel per GEMM in the trace
T = 22528
for name, m, n, k, ta, tb in (("gu_fwd NT", T, 22016, 4096, 0, 0), ("dh2 NN", T, 4096, 22016, 0, 1), ("gu_wgrad TT", 22016, 4096, T, 1, 1)):
    a = torch.randn((k, m) if ta else (m, k), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((k, n) if tb else (n, k), device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=c)
    torch.cuda.synchronize()
    del a, b, c

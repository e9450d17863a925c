"""Times the LoRA A-gradient (gA += dT^T dropout(x)) on the config-5 shapes: the one-pass kernel (rv_lora_a_grad_bf16) against the two-launch
sequence it replaced (rv_dropout_bf16, then the split-K weight-gradient GEMM).  Usage: python tools/lora_agrad_bench.py [out.json]"""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radvlm_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    out = {}
    ws = torch.empty(32 << 20, dtype=torch.float32, device="cuda")
    for M, K, r in [(22528, 5120, 64), (22528, 13824, 64), (22528, 4096, 64), (22528, 11008, 64)]:
        x = torch.randn(M, K, device="cuda").bfloat16()
        dt = torch.randn(M, r, device="cuda").bfloat16()
        ga = torch.zeros(r, K, device="cuda", dtype=torch.bfloat16)
        p, seed = 0.05, 1234
        t_new = timeit(lambda: ops.lora_a_grad(dt, x, ga, p, seed, True, ws))
        t_drop = timeit(lambda: ops.dropout(x, p, seed))
        xd = ops.dropout(x, p, seed)
        t_gemm = timeit(lambda: ops.gemm(dt, xd, ta=True, tb=True, out=ga, residual=ga, workspace=ws))
        gb = M * K * 2 / 1e9
        out[f"M{M}_K{K}_r{r}"] = {"one_pass_us": t_new, "dropout_us": t_drop, "split_k_gemm_us": t_gemm, "x_gbytes": gb,
                                 "one_pass_tb_per_s_over_x": gb / t_new * 1e3}
        print(M, K, r, f"one-pass {t_new:.1f} us ({gb / t_new * 1e3:.2f} TB/s over x) | dropout {t_drop:.1f} + gemm {t_gemm:.1f} us", flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()

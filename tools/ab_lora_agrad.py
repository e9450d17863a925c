"""Same-box A/B of the config-5 step (LoRA r = 64 on the 13B geometry): A = the A-gradients through the two-launch sequence of rounds 2-3
(rv_dropout_bf16, split-K weight-gradient GEMM), B = the one-pass kernel (rv_lora_a_grad_bf16).  The switch lives HERE (ops.lora_a_grad is
replaced for arm A), not in the product.  Usage: python tools/ab_lora_agrad.py A|B   (one arm per process; tools/ab_lora_agrad.sh alternates)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
arm = sys.argv[1]
from radvlm_amd import ops  # noqa: E402

if arm == "A":
    def two_launch(dt, x, ga, p, seed, accumulate, workspace):
        return ops.gemm(dt, ops.dropout(x, p, seed) if p > 0 else x, ta=True, tb=True, out=ga, residual=ga if accumulate else None, workspace=workspace)
    ops.lora_a_grad = two_launch
sys.argv = ["bench.py", "--workload", "lora", "--geometry", "llava15_13b", "--batch", "32", "--steps", "8", "--warmup", "3", "--no-cpu-baseline"]
import bench  # noqa: E402

bench.main()

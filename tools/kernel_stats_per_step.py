"""Per-step view of a `rocprofv3 --kernel-trace --stats` CSV of bench.py: total ms / steps per kernel family.

    python tools/kernel_stats_per_step.py profiles/r03_bench_b32_kernel_stats.csv 4 [out.json]

`steps` = the number of training steps the profiled command executed (bench.py --steps K --warmup W runs W + K timed steps + one
GEMM-instrumented step, + 2 more when a process group exists): the CSV holds whole-run totals, so every per-step figure quoted from it
in DESIGN.md is total / steps.  Start-up kernels (random init, casts, fills) are listed separately and not divided."""
import csv
import json
import re
import sys

FAMILIES = [("gemm_kernel_256", "GEMM 256x256 (all forms)"), ("gemm_nt_kernel", "GEMM 128x128"), ("tail_reduce", "GEMM tail reduce"),
            ("splitk_reduce", "GEMM split-K reduce"), ("attn_fwd_nat", "attention fwd (hd 128)"), ("attn_bwd_dq_nat", "attention bwd dQ"),
            ("attn_bwd_dkv_nat", "attention bwd dK/dV"), ("attn_fwd_kernel", "tower attention fwd (hd 64)"), ("attn_bwd_", "tower attention bwd"),
            ("adamw", "AdamW"), ("rmsnorm", "RMSNorm fwd+bwd"), ("layernorm", "LayerNorm"), ("colsum", "column sums"), ("sumsq", "grad norm"),
            ("cross_entropy", "cross entropy"), ("lora_down", "LoRA down-projection (dropout inside)"), ("lora_agrad", "LoRA A-gradient (mask re-created in registers)"), ("dropout", "dropout (LoRA)"), ("swiglu", "SwiGLU (unfused)"), ("rope", "RoPE (unfused)"),
            ("group_sum_heads", "GQA group sum")]
STARTUP = ("distribution_elementwise", "bfloat16_copy", "cast_bf16_f32", "copyBuffer", "FillFunctor<float>")


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    fam, startup, other = {}, 0.0, {}
    for r in rows:
        ms = float(r["TotalDurationNs"]) / 1e6
        name = r["Name"]
        if any(s in name for s in STARTUP):
            startup += ms
            continue
        for key, label in FAMILIES:
            if key in name:
                fam[label] = fam.get(label, 0.0) + ms
                break
        else:
            key = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", name.replace("(anonymous namespace)::", ""))[:60] or name[:60]
            other[key] = other.get(key, 0.0) + ms
    out = {"source": path, "steps_in_trace": steps, "ms_per_step": {k: round(v / steps, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])},
           "other_ms_per_step": round(sum(other.values()) / steps, 3),
           "other_largest": {k: round(v / steps, 3) for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:6]}, "startup_ms_total": round(startup, 2)}
    out["sum_ms_per_step"] = round(sum(out["ms_per_step"].values()) + out["other_ms_per_step"], 2)
    gemm = sum(v for k, v in out["ms_per_step"].items() if k.startswith("GEMM"))
    out["gemm_ms_per_step"] = round(gemm, 2)
    out["non_gemm_ms_per_step"] = round(out["sum_ms_per_step"] - gemm, 2)
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()

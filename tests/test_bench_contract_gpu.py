"""bench.py's one-line JSON contract (-m gpu): run it as the driver does, on the small config-1 geometry, and check the fields the
driver and the judge read -- also through the self-spawning `--gpus 2` form (rehearsal: both ranks on this one GPU over gloo, because
RCCL refuses two ranks per device; the measured N > 1 path uses nccl)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--geometry", "config1", "--batch", "4", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-1500:] + p.stderr[-1500:]
    # ONE line on stdout, nothing else (RCCL's version banner, printed when the first communicator comes up, must not land there)
    assert [l for l in p.stdout.splitlines() if l.strip()] == lines, p.stdout[:1500]
    return json.loads(lines[0])


def _check(d, n):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "distributed", "build"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["global_batch"] == 4 * n
    assert abs(d["value"] - 4 * n / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]          # whole-job pairs over the timed steps
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None and "no PMC pass on record" in r["traffic_source"]         # PMC numbers are quoted for the headline workload only
    assert r["algorithmic_bytes_per_launch"] > 0 and r["launches_per_step"] > 0
    assert d["distributed"]["world"] == n and d["distributed"]["ranks_seen"] == n
    assert len(d["build"]["kernel_source_sha256"]) == 16
    assert 3.0 < d["config"]["final_loss"] < 12.5


def test_single_gpu_line():
    d = _run([])
    _check(d, 1)
    assert d["distributed"]["backend"] is None and d["distributed"]["exposed_comm_ms_per_step"] is None


def test_self_spawned_two_rank_line():
    d = _run(["--gpus", "2"], env={"RV_BENCH_REHEARSAL": "1"})
    _check(d, 2)
    assert "gloo" in d["distributed"]["backend"] and d["distributed"]["exposed_comm_ms_per_step"] >= 0.0
    # one driver run at N > 1 tells exposed communication, straggler spread over ranks and how many collectives a step issues
    dd = d["distributed"]
    assert 0.0 < dd["ms_per_step_min"] <= dd["ms_per_step_max"] <= d["ms_per_step"] * 1.001 and dd["bucket_count"] >= 2


def test_rccl_branch_runs_in_a_one_rank_group():
    """The `nccl` (= RCCL) branch on the one GPU a test box has: `init_process_group("nccl", world_size=1)` before any other GPU call,
    every gradient bucket's all-reduce issued on the side stream next to backward's GEMMs (one-tile GEMM blocks,
    rv_gemm_select_kernel(40), as for N > 1), the compute stream joined with it before the optimizer.  A one-rank all-reduce leaves the
    gradients unchanged, so the loss trajectory must equal the plain single-GPU run's bit for bit."""
    d0 = _run([])
    d = _run(["--force-process-group"])
    _check(d, 1)
    assert d["distributed"]["backend"] == "nccl" and d["distributed"]["exposed_comm_ms_per_step"] >= 0.0
    assert d["distributed"]["grad_sync"] is not None
    assert d["config"]["final_loss"] == d0["config"]["final_loss"], (d["config"]["final_loss"], d0["config"]["final_loss"])

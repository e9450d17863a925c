#!/bin/bash
# Samples the GPU's package power, power cap and engine clock (rocm-smi, read-only) every 0.2 s while the headline bench runs, and prints the
# distribution over the timed steps.  Evidence for DESIGN.md section 5 ("the GEMM step runs at the board's power cap, not at 2.4 GHz").
# Usage (on the GPU box, through gpurun): bash tools/power_probe.sh > gpurun_out/r04_power_clock_during_bench.txt
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/power_probe_samples.txt
: > $O
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max|cap" | head -3
( while true; do echo "T $(date +%s.%N)" >> $O; rocm-smi -P -c -t 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)" >> $O; sleep 0.2; done ) &
SAMPLER=$!
sleep 1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/power_probe_bench.json 2>/dev/null
kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
python3 - <<'PY'
import json, re
d = json.load(open("gpurun_out/power_probe_bench.json"))
print("bench: %.2f pairs/s, %.1f ms/step, GEMM frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
pw, ck = [], []
for line in open("gpurun_out/power_probe_samples.txt"):
    m = re.search(r"Power.*?:\s*([0-9.]+)", line)
    if m: pw.append(float(m.group(1)))
    m = re.search(r"sclk.*?\((\d+)Mhz\)", line)
    if m: ck.append(int(m.group(1)))
def dist(v):
    v = sorted(v); n = len(v)
    return "n=%d min %.0f p25 %.0f median %.0f p75 %.0f max %.0f" % (n, v[0], v[n // 4], v[n // 2], v[3 * n // 4], v[-1]) if v else "none"
print("package power [W]:", dist(pw))
print("sclk [MHz]       :", dist(ck))
# the busy part of the run: samples above half of the maximum power
if pw:
    hi = [p for p in pw if p > 0.5 * max(pw)]
    print("package power while the step runs [W]:", dist(hi))
PY
echo "--- raw sample (one) ---"; grep -A6 "^T" $O | sed -n 1,8p

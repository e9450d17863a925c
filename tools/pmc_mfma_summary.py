"""Aggregate a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_BUSY_CYCLES --kernel-trace` counter_collection.csv per kernel family:
  clock      = GRBM_GUI_ACTIVE / 8 XCDs / dispatch wall time           (MI355X_MICROARCH.md, DVFS give-back)
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)   (fraction of MFMA-pipe cycles busy at the held clock)
  wave-cycle split: ACTIVE_INST_ANY / WAIT_INST_ANY / WAIT_ANY over WAVE_CYCLES (disjoint buckets)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from radvlm_amd.build_id import kernel_source_sha256  # noqa: E402

path, out = sys.argv[1], sys.argv[2]
disp = defaultdict(dict)
name = {}
gridsz = {}
with open(path) as f:
    for r in csv.DictReader(f):
        d = r["Dispatch_Id"]
        disp[d][r["Counter_Name"]] = float(r["Counter_Value"])
        disp[d]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        name[d] = r["Kernel_Name"]
        gridsz[d] = r["Grid_Size"]
fam = defaultdict(lambda: defaultdict(float))
for d, c in disp.items():
    n = name[d]
    m = re.search(r"(gemm_kernel_256<[^>]*>|gemm_nt_kernel|attn_\w+_kernel<[^>]*>|adamw_kernel|swiglu_\w+_kernel|rmsnorm_\w+_kernel)", n)
    if m and m.group(1).startswith("attn_"):        # attention: keep the launch shapes apart (grid size tells the sequence length)
        m = re.match(r"(.*)", m.group(1) + " grid=" + str(gridsz[d]))
    if not m:
        continue
    k = m.group(1)
    for key, v in c.items():
        fam[k][key] += v
    fam[k]["launches"] += 1
res = {"source": "rocprofv3 --pmc (own pass, --kernel-trace only) on `python bench.py --steps 1 --warmup 1 --no-cpu-baseline` (b=32), MI355X",
       "kernel_source_sha256": kernel_source_sha256(), "git_commit": sys.argv[3] if len(sys.argv) > 3 else None, "kernels": {}}
tot = defaultdict(float)
for k, c in sorted(fam.items(), key=lambda kv: -kv[1]["_ns"]):
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    e = dict(launches=int(c["launches"]), total_ms=c["_ns"] / 1e6, clock_ghz=cyc / c["_ns"],
             mfma_busy_frac=c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
             wave_active_frac=c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], wave_issue_stall_frac=c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
             wave_parked_frac=c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"])
    res["kernels"][k] = e
    if k.startswith("gemm_kernel_256"):
        for key in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "_ns", "launches"):
            tot[key] += c[key]
cyc = tot["GRBM_GUI_ACTIVE"] / 8.0
if tot["_ns"] > 0:
    res["gemm_kernel_256_all_forms"] = dict(launches=int(tot["launches"]), total_ms=tot["_ns"] / 1e6, clock_ghz=cyc / tot["_ns"],
                                            mfma_busy_frac=tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
                                            mfma_busy_frac_of_2p4ghz=tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (tot["_ns"] * 2.4 * 1024.0))
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res.get("gemm_kernel_256_all_forms")))
for k, e in res["kernels"].items():
    print(f"{k:48s} n={e['launches']:5d} {e['total_ms']:8.1f} ms clk {e['clock_ghz']:.2f} GHz mfma {e['mfma_busy_frac']:.3f} "
          f"active {e['wave_active_frac']:.2f} issue-stall {e['wave_issue_stall_frac']:.2f} parked {e['wave_parked_frac']:.2f}")

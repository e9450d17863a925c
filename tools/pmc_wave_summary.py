"""Per-kernel wave-state view of a `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv` counter_collection.csv, attention kernels only:
  parked (s_waitcnt / barrier), issue-stalled, issuing -- fractions of the waves' cycles; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x cycles)."""
import csv, json, re, sys
from collections import defaultdict
disp, name, grid = defaultdict(dict), {}, {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        d = r["Dispatch_Id"]
        disp[d][r["Counter_Name"]] = float(r["Counter_Value"])
        disp[d]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        name[d], grid[d] = r["Kernel_Name"], r["Grid_Size"]
fam = defaultdict(lambda: defaultdict(float))
for d, c in disp.items():
    m = re.search(r"(attn_\w+_kernel<[^>]*>)", name[d])
    if not m:
        continue
    k = m.group(1) + " grid=" + grid[d]
    for key, v in c.items():
        fam[k][key] += v
    fam[k]["n"] += 1
out = {}
for k, c in sorted(fam.items(), key=lambda kv: -kv[1]["_ns"]):
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    w = max(c["SQ_WAVE_CYCLES"], 1.0)
    e = out[k] = dict(launches=int(c["n"]), avg_us=c["_ns"] / c["n"] / 1e3, clock_ghz=cyc / c["_ns"], mfma_pipe_busy=c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
                      wave_parked=c["SQ_WAIT_ANY"] / w, wave_issue_stall=c["SQ_WAIT_INST_ANY"] / w, wave_active=c["SQ_ACTIVE_INST_ANY"] / w,
                      valu_active=c.get("SQ_ACTIVE_INST_VALU", 0.0) / w)
    print(f"{k:64s} n={e['launches']:3d} {e['avg_us']:8.1f} us  clock {e['clock_ghz']:.2f} GHz  MFMA busy {e['mfma_pipe_busy']:.3f}  waves: parked {e['wave_parked']:.2f} "
          f"issue-stall {e['wave_issue_stall']:.2f} active {e['wave_active']:.2f} (VALU {e['valu_active']:.2f})")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)

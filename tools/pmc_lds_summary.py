"""Per-kernel LDS / wave-state view of a `rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace` counter_collection.csv (own pass; 8 SQ slots on gfx950):
  lds_busy   = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 * 256 CUs)   fraction of the LDS arrays' cycles spent serving reads / writes
  conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE               extra cycles lost to bank conflicts
  wave cycles: waiting to issue an LDS instruction (SQ_WAIT_INST_LDS, a sub-bucket of issue-stall), issue-stall, parked, active."""
import csv
import json
import re
import sys
from collections import defaultdict

path = sys.argv[1]
disp, name, grid = defaultdict(dict), {}, {}
with open(path) as f:
    for r in csv.DictReader(f):
        d = r["Dispatch_Id"]
        disp[d][r["Counter_Name"]] = float(r["Counter_Value"])
        disp[d]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        name[d], grid[d] = r["Kernel_Name"], r["Grid_Size"]
fam = defaultdict(lambda: defaultdict(float))
for d, c in disp.items():
    m = re.search(r"(gemm_kernel_256<[^>]*>|attn_\w+_kernel<[^>]*>)", name[d])
    if not m:
        continue
    k = m.group(1) + (" grid=" + grid[d] if m.group(1).startswith("attn_") else "")
    for key, v in c.items():
        fam[k][key] += v
    fam[k]["n"] += 1
out = {}
for k, c in sorted(fam.items(), key=lambda kv: -kv[1]["_ns"]):
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    w = max(c["SQ_WAVE_CYCLES"], 1.0)
    out[k] = dict(launches=int(c["n"]), total_ms=c["_ns"] / 1e6, lds_busy_frac=c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0),
                  lds_conflict_frac=c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), wave_wait_lds_issue=c["SQ_WAIT_INST_LDS"] / w,
                  wave_issue_stall=c["SQ_WAIT_INST_ANY"] / w, wave_parked=c["SQ_WAIT_ANY"] / w, wave_active=c["SQ_ACTIVE_INST_ANY"] / w)
    e = out[k]
    print(f"{k:60s} n={e['launches']:4d} {e['total_ms']:8.2f} ms  lds busy {e['lds_busy_frac']:.2f} conflict {e['lds_conflict_frac']:.3f}  "
          f"waves: lds-issue {e['wave_wait_lds_issue']:.2f} issue-stall {e['wave_issue_stall']:.2f} parked {e['wave_parked']:.2f} active {e['wave_active']:.2f}")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)

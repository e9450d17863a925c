"""Summarise the tile-order probes: per GROUP_M variant and GEMM, L2-miss bytes (FETCH_SIZE x 2, gfx950 correction) and time."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

out = {}
for g in sys.argv[2:]:
    files = glob.glob(f"gpurun_out/gm{g}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    rows = defaultdict(lambda: [0.0, 0.0, 0])
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            m = re.search(r"gemm_kernel_256<(\w+), (\w+),", r["Kernel_Name"])
            if not m or r["Counter_Name"] != "FETCH_SIZE":
                continue
            k = {"false, false": "gu_fwd NT", "false, true": "dh2 NN", "true, true": "gu_wgrad TT"}[f"{m.group(1)}, {m.group(2)}"]
            rows[k][0] += float(r["Counter_Value"]) * 1024 * 2
            rows[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            rows[k][2] += 1
    out[f"GROUP_M={g}"] = {k: {"l2_miss_GB_per_launch": v[0] / v[2] / 1e9, "us_per_launch_under_pmc": v[1] / v[2] / 1e3} for k, v in rows.items()}
alg = {"gu_fwd NT": (22528 * 4096 + 22016 * 4096 + 22528 * 22016) * 2 / 1e9, "dh2 NN": (22528 * 22016 + 22016 * 4096 + 22528 * 4096) * 2 / 1e9,
       "gu_wgrad TT": (22528 * 22016 + 22528 * 4096 + 22016 * 4096) * 2 / 1e9}
res = {"what": "block -> tile order of the 256x256 GEMM: GROUP_M tile rows per group (88 = whole tile columns), rocprofv3 --pmc FETCH_SIZE on "
               "tools/tile_order_probe.py (3 launches per shape, plain launch shape)", "algorithmic_GB": alg, "variants": out}
json.dump(res, open(sys.argv[1], "w"), indent=1)
for v, d in out.items():
    print(v, {k: (round(x["l2_miss_GB_per_launch"], 2), round(x["us_per_launch_under_pmc"])) for k, x in d.items()})
print("algorithmic GB", {k: round(v, 2) for k, v in alg.items()})

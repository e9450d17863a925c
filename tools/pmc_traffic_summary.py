"""HBM-side bytes per GEMM launch from two separate rocprofv3 passes (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each with
--kernel-trace only) of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`.

    python tools/pmc_traffic_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json [git-commit]

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counter unit 1 KiB; on gfx950 FETCH_SIZE reports half the bytes
of wide coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Calibrated in the same run on
adamw_kernel, whose algorithmic traffic is known exactly (per parameter: reads 2 + 3*4 = 14 B, writes 2 + 3*4 = 14 B).
The JSON carries the hash of the kernel sources the run was made with (radvlm_amd.build_id): bench.py only quotes it when that hash
matches the sources it runs."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from radvlm_amd.build_id import kernel_source_sha256  # noqa: E402


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(gemm_kernel_256<[^>]*>|gemm_nt_kernel|tail_reduce_kernel|splitk_reduce_kernel|adamw_kernel|attn_\w+_kernel)", r["Kernel_Name"])
            if not m:
                continue
            tot[m.group(1)] += float(r["Counter_Value"]) * 1024.0
            n[m.group(1)] += 1
    return tot, n


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python3 bench.py --steps 1 --warmup 1 "
                 "--no-cpu-baseline` (b=32), MI355X",
       "correction": "bytes = counter * 1024; FETCH_SIZE doubled (gfx950 counts half of wide coalesced reads); WRITE_SIZE as is",
       "kernel_source_sha256": kernel_source_sha256(), "git_commit": sys.argv[4] if len(sys.argv) > 4 else None, "per_operand_form": {}}
if "adamw_kernel" in fetch:
    res["calibration_adamw"] = {"fetch_x2_bytes": 2 * fetch["adamw_kernel"], "write_bytes": write.get("adamw_kernel"),
                                "note": "algorithmic: 14 B read + 14 B written per parameter element of the launches' slices"}
R = W = L = 0.0
for k in fetch:
    if k.startswith("gemm_kernel_256") or k in ("tail_reduce_kernel", "splitk_reduce_kernel"):
        R += 2 * fetch[k]
        W += write.get(k, 0.0)
        if k.startswith("gemm_kernel_256"):
            L += nf[k]
            res["per_operand_form"][k] = {"launches": nf[k], "read_per_launch": 2 * fetch[k] / nf[k], "write_per_launch": write.get(k, 0.0) / max(1, nw.get(k, 1))}
res["gemm_kernel_256"] = {"launches": int(L), "hbm_read_bytes_per_launch": R / L, "hbm_write_bytes_per_launch": W / L, "hbm_bytes_per_launch": (R + W) / L,
                          "note": "reduce kernels of the tail-split / split-K launch shapes are charged to their GEMM launch"}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res["gemm_kernel_256"]))

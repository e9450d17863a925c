#!/bin/bash
# Memory-side counters of the attention kernels on the micro-benchmark (run through gpurun): L2-miss traffic (FETCH_SIZE / WRITE_SIZE, separate
# passes as the guide prescribes) and L2 hit / miss / request counts.  Usage: bash tools/pmc_attn_memory.sh > gpurun_out/r04_pmc_attn_memory.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_attn
rm -rf $O; mkdir -p $O
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$tag -- python3 tools/attn_bench.py > $O/$tag.log 2>&1 || { tail -3 $O/$tag.log; continue; }
done
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_attn/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_" not in k or "nat" not in k: continue
        key = (k.split("(")[0][-40:], r["Grid_Size"])
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(rows):
    d = rows[key]
    out = {c: sum(v) / len(v) for c, v in d.items()}
    s = f"{key[0]:42s} grid {key[1]:>9s}: "
    if "FETCH_SIZE" in out: s += f"L2-miss reads {out['FETCH_SIZE'] * 1024 * 2 / 1e6:8.0f} MB (FETCH_SIZE x 1024 x 2)  "
    if "WRITE_SIZE" in out: s += f"writes {out['WRITE_SIZE'] * 1024 / 1e6:7.0f} MB  "
    if "TCC_HIT_sum" in out: s += f"L2 hit {out['TCC_HIT_sum'] / max(1, out['TCC_HIT_sum'] + out['TCC_MISS_sum']):.3f} of {out['TCC_HIT_sum'] + out['TCC_MISS_sum']:.3g} lookups  "
    if "TCC_REQ_sum" in out: s += f"L2 requests {out['TCC_REQ_sum']:.3g}  "
    if "TCC_EA0_RDREQ_sum" in out: s += f"EA read requests {out['TCC_EA0_RDREQ_sum']:.3g}"
    print(s)
PY
rm -rf $O

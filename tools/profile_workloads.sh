#!/bin/bash
# Kernel traces of the non-headline workloads on the GPU box (run through gpurun): rocprofv3 --kernel-trace --stats of bench.py itself
# (program directly behind `--`), summaries copied to gpurun_out/<tag>_<workload>_kernel_stats.csv.   Usage: bash tools/profile_workloads.sh r03 [workloads...]
set -u
TAG=${1:-r03}; shift || true
WL=${@:-"lora anyres radvlm"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for w in $WL; do
  case $w in
    lora)   ARGS="--workload lora --geometry llava15_13b --batch 32" ;;
    anyres) ARGS="--workload anyres --batch 8" ;;
    radvlm) ARGS="--workload radvlm --geometry llava_ov_qwen2_7b --batch 2" ;;
    cxr)    ARGS="" ;;
    *) echo "unknown workload $w"; exit 2 ;;
  esac
  echo "[profile] $w"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_$w -- python3 bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_trace_$w.log 2>&1 || { tail -5 $O/${TAG}_trace_$w.log; exit 1; }
  cp "$(find $O/${TAG}_trace_$w -name '*kernel_stats.csv' | head -1)" $O/${TAG}_${w}_kernel_stats.csv
  grep '^{' $O/${TAG}_trace_$w.log > $O/${TAG}_${w}_profiled_bench_line.json || true
  rm -rf $O/${TAG}_trace_$w
  head -12 $O/${TAG}_${w}_kernel_stats.csv | cut -c1-160
done
echo "[profile] done"

#!/bin/bash
# One measurement session on the GPU box (run through gpurun): bench line, kernel trace, the three PMC passes of the same command and the
# summaries bench.py quotes (profiles/README.md).  Usage: bash tools/measure_round.sh r02   -> files under gpurun_out/<tag>_*
# Every rocprofv3 run profiles `python3 bench.py ...` directly (no shell / env hop between the profiler and the program).
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B="--steps 1 --warmup 1 --no-cpu-baseline"
echo "[measure] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_trace.log 2>&1 || exit 1
cp "$(find $O/${TAG}_trace -name '*kernel_stats.csv' | head -1)" $O/${TAG}_bench_b32_kernel_stats.csv
echo "[measure] pmc FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -- python3 bench.py $B > $O/${TAG}_pmc_fetch.log 2>&1 || exit 1
echo "[measure] pmc WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -- python3 bench.py $B > $O/${TAG}_pmc_write.log 2>&1 || exit 1
echo "[measure] pmc MFMA"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/${TAG}_pmc_mfma -- python3 bench.py $B > $O/${TAG}_pmc_mfma.log 2>&1 || exit 1
python3 tools/pmc_traffic_summary.py "$(find $O/${TAG}_pmc_fetch -name '*counter_collection.csv' | head -1)" "$(find $O/${TAG}_pmc_write -name '*counter_collection.csv' | head -1)" $O/${TAG}_pmc_gemm_hbm_traffic.json
python3 tools/pmc_mfma_summary.py "$(find $O/${TAG}_pmc_mfma -name '*counter_collection.csv' | head -1)" $O/${TAG}_pmc_mfma_util.json | tail -25
# keep the merged-back payload small: the raw per-dispatch CSVs stay on the box
rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_mfma $O/${TAG}_trace
echo "[measure] done"

#!/bin/bash
# Re-captures the judged evidence under gpurun_out/ (copy the summaries into profiles/ afterwards).
# usage (on the GPU box, from the repo root):  bash tools/capture_profiles.sh r01
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/${TAG}_bench_n1.json" 2> "$OUT/bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/mfma" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline > "$OUT/mfma.log" 2>&1
cd "$REPO"
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
M=$(find "$OUT/mfma" -name '*counter_collection.csv' | head -1)
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 tools/summarize_pmc.py "$F" "$W" > "$OUT/${TAG}_pmc_hbm_traffic.csv"
python3 tools/summarize_mfma.py "$M" > "$OUT/${TAG}_pmc_mfma_utilisation.csv" 2> "$OUT/mfma_sum.err"
cp "$S" "$OUT/${TAG}_kernel_stats_full.csv"
# keep the traces themselves out of the merged gpurun_out (size)
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/mfma"
ls -la "$OUT"

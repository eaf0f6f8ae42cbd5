#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_al; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/tools/ubench/train_step.py" --batch 64 --frames 1000 --steps 3 > "$OUT/train.log" 2>&1; echo rc=$?
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 "$REPO/tools/summarize_kernels.py" "$S" > "$OUT/r02_kernel_stats_train_step.csv" 2>/dev/null
head -25 "$OUT/r02_kernel_stats_train_step.csv" | cut -c1-150
rm -rf "$OUT/stats"

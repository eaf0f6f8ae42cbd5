#!/bin/bash
# Kernel-level profile of the training step as it stands at the end of round 2 (84 ms at 64 x 1000); the "before" is r02_al.sh.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r02_am"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/tools/ubench/train_step.py" --batch 64 --frames 1000 --steps 3 > "$OUT/train.log" 2>&1; echo rc=$?
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 "$REPO/tools/summarize_kernels.py" "$S" > "$OUT/r02_kernel_stats_train_step_final.csv" 2>/dev/null
head -30 "$OUT/r02_kernel_stats_train_step_final.csv" | cut -c1-170
tail -2 "$OUT/train.log"
rm -rf "$OUT/stats"

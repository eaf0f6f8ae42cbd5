#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_o; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
for b in 8 16; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_b$b" -- python3 "$REPO/tools/ubench/small_batch_trace.py" --batch $b > "$OUT/run_b$b.log" 2>&1; echo rc=$?; tail -2 "$OUT/run_b$b.log"
f=$(find "$OUT/trace_b$b" -name '*kernel_trace.csv' | head -1)
python3 "$REPO/tools/summarize_gaps.py" "$f" --skip-first 0.6 > "$OUT/gaps_b$b.txt"; cat "$OUT/gaps_b$b.txt"
rm -rf "$OUT/trace_b$b"
done

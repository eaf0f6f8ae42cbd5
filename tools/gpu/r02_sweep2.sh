#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_sweep2; mkdir -p "$OUT/sweep"; export TMPDIR=/tmp
timeout 2400 python tools/latency_sweep.py --out "$OUT/sweep" --summary "$OUT/sweep/r02_latency_sweep_summary.json" > "$OUT/sweep.log" 2>&1; echo "sweep rc=$?"; tail -4 "$OUT/sweep.log"
ls -la "$OUT/sweep"

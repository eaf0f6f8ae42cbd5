#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_ai; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python tools/ubench/ab_gc_osplit.py --batches 64 --lnx0 > "$OUT/ab_lnx0.log" 2>&1; echo rc=$?
grep '"lnx"' "$OUT/ab_lnx0.log"

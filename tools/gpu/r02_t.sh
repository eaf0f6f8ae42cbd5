#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_t; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bf16_ops_gpu.py -x -q -k "variants or ragged or output_split" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
timeout 900 python tools/ubench/ab_gc_osplit.py > "$OUT/ab_osplit.log" 2>&1; echo rc=$?; tail -26 "$OUT/ab_osplit.log"

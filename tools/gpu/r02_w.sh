#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_w; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bf16_ops_gpu.py -x -q -k "variants or ragged or output_split" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
timeout 900 python tools/ubench/gc_phases.py > "$OUT/gc_phases.log" 2>&1; echo rc=$?; tail -5 "$OUT/gc_phases.log"
timeout 900 python tools/ubench/ab_gc_osplit.py --batches 64 > "$OUT/ab_pipe.log" 2>&1; echo rc=$?; tail -13 "$OUT/ab_pipe.log"

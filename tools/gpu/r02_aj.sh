#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_aj; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_backward_gpu.py -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -25 "$OUT/pytest.log"

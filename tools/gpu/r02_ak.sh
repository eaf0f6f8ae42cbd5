#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_ak; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_backward_gpu.py -q -s -k "sgd_step" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -30 "$OUT/pytest.log"

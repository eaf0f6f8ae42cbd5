#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_ak; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_backward_gpu.py -q -s -k "loss_backward" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -30 "$OUT/pytest.log"
timeout 600 python -m pytest tests/test_backward_gpu.py -q > "$OUT/pytest_bw.log" 2>&1; echo "bw rc=$?"; tail -3 "$OUT/pytest_bw.log"
timeout 600 python tools/ubench/train_step.py --batch 64 --frames 1000 --steps 2 2>/dev/null | tail -1

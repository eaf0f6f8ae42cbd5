#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r02_ao"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$REPO"
timeout 600 python -m pytest tests/test_backward_gpu.py -q -x > "$OUT/pytest_bw.log" 2>&1; echo "bw rc=$?"; tail -15 "$OUT/pytest_bw.log"
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "loss_backward or sgd_step" > "$OUT/pytest_model.log" 2>&1; echo "model rc=$?"; tail -3 "$OUT/pytest_model.log"
timeout 600 python tools/ubench/train_step.py --batch 64 --frames 1000 --steps 3 2>/dev/null | tail -1
timeout 600 python tools/ubench/train_step.py --batch 16 --frames 400 --steps 3 2>/dev/null | tail -1

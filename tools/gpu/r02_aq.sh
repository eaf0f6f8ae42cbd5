#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r02_aq"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$REPO"
timeout 600 python -m pytest tests/test_backward_gpu.py -q -x > "$OUT/pytest_bw.log" 2>&1; echo "bw rc=$?"; tail -2 "$OUT/pytest_bw.log"
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "loss_backward or sgd_step or training_mode" > "$OUT/pytest_model.log" 2>&1; echo "model rc=$?"; tail -2 "$OUT/pytest_model.log"; grep "parameter gradients checked" "$OUT/pytest_model.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
{ timeout 600 python tools/ubench/train_step.py --batch 16 --frames 400 --steps 3 2>/dev/null | tail -1
  timeout 600 python tools/ubench/train_step.py --batch 64 --frames 1000 --steps 3 2>/dev/null | tail -1; } > "$OUT/r02_train_step_final.jsonl"
cat "$OUT/r02_train_step_final.jsonl"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/tools/ubench/train_step.py" --batch 64 --frames 1000 --steps 3 > "$OUT/train.log" 2>&1; echo "train profile rc=$?"
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 "$REPO/tools/summarize_kernels.py" "$S" > "$OUT/r02_kernel_stats_train_step_final.csv" 2>/dev/null
rm -rf "$OUT/stats"
head -8 "$OUT/r02_kernel_stats_train_step_final.csv" | cut -c1-140

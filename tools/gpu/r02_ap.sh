#!/bin/bash
# Final capture of round 2: judged evidence (bench lines, kernel stats, counters), the training step (timing + kernel profile), full GPU suite.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_ap; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 bash tools/capture_profiles.sh r02 > "$OUT/capture.log" 2>&1; echo "capture rc=$?"
cd "$REPO"
{ timeout 600 python tools/ubench/train_step.py --batch 16 --frames 400 --steps 3 2>/dev/null | tail -1
  timeout 600 python tools/ubench/train_step.py --batch 64 --frames 1000 --steps 3 2>/dev/null | tail -1; } > "$OUT/r02_train_step_final.jsonl"
cat "$OUT/r02_train_step_final.jsonl"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/tools/ubench/train_step.py" --batch 64 --frames 1000 --steps 3 > "$OUT/train.log" 2>&1; echo "train profile rc=$?"
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 "$REPO/tools/summarize_kernels.py" "$S" > "$OUT/r02_kernel_stats_train_step_final.csv" 2>/dev/null
rm -rf "$OUT/stats"
cd "$REPO"
timeout 2400 python -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
python - <<'PY'
import json
for f in ('r02_bench_n1.json', 'r02_bench_cfg3_bf16.json'):
    try:
        d = json.loads(open('gpurun_out/profiles_r02/' + f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d.get('config', {}).get('build_id'))
    except Exception as e:
        print(f, 'parse failed', e)
PY

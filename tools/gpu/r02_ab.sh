#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_ab; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" "$OUT/pytest.log" | tail -2; grep -E "^FAILED" "$OUT/pytest.log" | head
for rep in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-strict 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B64', round(d['value']), round(d['ms_per_step'], 3), round(d['roofline']['frac'], 4), round(d['value_sequential']), d['p50_forward_ms'], d['ms_per_forward_by_kernel'])"
done
for b in 8 16 32; do timeout 300 python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --no-strict --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', $b, round(d['value']), round(d['ms_per_step'], 3))"; done

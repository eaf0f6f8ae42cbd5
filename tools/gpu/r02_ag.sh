#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_ag; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "dense or first_conv" > "$OUT/pytest_ops.log" 2>&1; echo "ops rc=$?"; tail -3 "$OUT/pytest_ops.log"
timeout 2400 python -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" "$OUT/pytest.log" | tail -2; grep -E "^FAILED" "$OUT/pytest.log" | head
for rep in 1 2 3; do
for mode in 1 0; do
NBASR_DENSE_EPILOGUE_STATS=$mode timeout 600 python bench.py --no-cpu-baseline --no-strict 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['ms_per_forward_by_kernel']; print('dense stats', $mode, round(d['value']), round(d['ms_per_step'], 3), round(k['dense_conv'], 3), round(k.get('channel_stats', 0), 3), round(k['stats_finalize'], 3))"
done; done

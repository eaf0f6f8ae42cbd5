#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_n; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tape_gpu.py -x -q > "$OUT/pytest_tape.log" 2>&1; echo "tape tests rc=$?"; tail -15 "$OUT/pytest_tape.log"
timeout 600 python tools/ubench/host_issue.py > "$OUT/host_issue_tape.log" 2>&1; echo rc=$?; tail -5 "$OUT/host_issue_tape.log"
NBASR_TAPE=0 timeout 600 python tools/ubench/host_issue.py --batches 8 64 > "$OUT/host_issue_notape.log" 2>&1; tail -3 "$OUT/host_issue_notape.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo bench rc=$?; python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r02_n/bench.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['roofline']['frac'])
except Exception as e:
    print('bench parse failed', e)
PY
for b in 8 16 32; do timeout 300 python bench.py --batch $b --steps 40 --warmup 8 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', $b, d['value'], d['ms_per_step'])"; done

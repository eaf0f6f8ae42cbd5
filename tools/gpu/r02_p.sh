#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_p; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -12 "$OUT/pytest.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo bench rc=$?; python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r02_p/bench.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('roofline_mfma', {}).get('per_layer'))
    print({k: v for k, v in d.items() if k in ('latency_ms', 'by_kind_ms')})
except Exception as e:
    print('bench parse failed', e)
PY
for b in 8 16 32; do timeout 300 python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --no-strict 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', $b, d['value'], d['ms_per_step'])"; done

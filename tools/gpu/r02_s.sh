#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_s; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 bash tools/capture_profiles.sh r02 > "$OUT/capture.log" 2>&1; echo "capture rc=$?"
cd "$REPO"
timeout 2400 python -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
python - <<'PY'
import json
for f in ('r02_bench_n1.json', 'r02_bench_cfg3_bf16.json'):
    try:
        d = json.loads(open('gpurun_out/profiles_r02/' + f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))
    except Exception as e:
        print(f, 'parse failed', e)
PY

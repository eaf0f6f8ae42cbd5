#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_aa; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python tools/ubench/ab_gc_osplit.py --batches 64 8 --deep > "$OUT/ab_deep_k5d1.log" 2>&1; echo rc=$?
timeout 900 python tools/ubench/ab_gc_osplit.py --batches 64 --kernel 7 --dilation 2 --deep > "$OUT/ab_deep_k7d2.log" 2>&1; echo rc=$?
python - <<'PY'
import json
for f in ('ab_deep_k5d1.log', 'ab_deep_k7d2.log'):
    for line in open('gpurun_out/r02_aa/' + f):
        if line.startswith('{'):
            d = json.loads(line)
            print(d['k'], d['d'], 'B', d['batch'], 'blk', d['block'], d['flavour'], d['skips'], '|', d['default_us'], d['osplit_us'], d['pipe_us'], d['pipe_osplit_us'], '| deep', d['deep_us'], d['deep_osplit_us'])
PY

#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_x; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python tools/ubench/ab_gc_osplit.py --batches 64 8 --kernel 7 --dilation 2 > "$OUT/ab_pipe_k7d2.log" 2>&1; echo rc=$?
timeout 900 python tools/ubench/ab_gc_osplit.py --batches 8 32 > "$OUT/ab_pipe_k5_small.log" 2>&1; echo rc=$?
python - <<'PY'
import json
for f in ('ab_pipe_k7d2.log', 'ab_pipe_k5_small.log'):
    for line in open('gpurun_out/r02_x/' + f):
        if line.startswith('{'):
            d = json.loads(line)
            print(d['k'], d['d'], 'B', d['batch'], 'blk', d['block'], d['flavour'], d['skips'], '|', d['default_us'], d['osplit_us'], d['pipe_us'], d['pipe_osplit_us'])
PY

#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_ae; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "dense or f16 or image or packed or first_conv" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-strict --no-roofline > "$OUT/stats.log" 2>&1
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 - "$S" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('gemm_conv_split', )):
        print(n.split('(')[0][:90].ljust(90), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f} us")
PY
tail -1 "$OUT/stats.log" | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])"
rm -rf "$OUT/stats"

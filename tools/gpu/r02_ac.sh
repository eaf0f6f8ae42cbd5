#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_ac; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_bf16_ops_gpu.py -q -x -k "layernorm or stats or image or first_conv" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-strict --no-roofline > "$OUT/stats.log" 2>&1
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 - "$S" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('channel_stats', 'normalize_split', 'layernorm_channels', 'pw_', 'stats_finalize', 'input_range')):
        print(n.split('(')[0][:60].ljust(60), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f} us")
PY
rm -rf "$OUT/stats"

#!/bin/bash
# round 2, GPU call D: whole -m gpu suite, small-batch lanes A/B, then the full 8 242-architecture latency sweep (config 5)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_d
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" "$OUT/pytest.log" | tail -3
grep -E "^FAILED|^ERROR|exact-fp32 mode" "$OUT/pytest.log" | head -40
for lanes in 0 2 3; do
  NBASR_GRAPH_LANES=$lanes timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-strict --no-roofline --steps 60 --warmup 10 > "$OUT/b8_lanes$lanes.json" 2> "$OUT/b8_lanes$lanes.err"
  python3 -c "import json,sys; d=json.loads(open('$OUT/b8_lanes$lanes.json').read().strip().splitlines()[-1]); print('B=8 lanes $lanes', round(d['value']), 'utt/s', round(d['ms_per_step'],3), 'ms/step; sequential', round(d['value_sequential']), 'p50', round(d['p50_forward_ms'],3))" || tail -3 "$OUT/b8_lanes$lanes.err"
done
for b in 16 32; do
for lanes in 0 2; do
  NBASR_GRAPH_LANES=$lanes timeout 300 python bench.py --batch $b --no-cpu-baseline --no-strict --no-roofline --steps 40 --warmup 10 > "$OUT/b${b}_lanes$lanes.json" 2> /dev/null
  python3 -c "import json,sys; d=json.loads(open('$OUT/b${b}_lanes$lanes.json').read().strip().splitlines()[-1]); print('B=$b lanes $lanes', round(d['value']), 'utt/s', round(d['ms_per_step'],3), 'ms/step')"
done; done
mkdir -p "$OUT/sweep"
timeout 2400 python tools/latency_sweep.py --out "$OUT/sweep" --summary "$OUT/sweep/r02_latency_sweep_summary.json" > "$OUT/sweep.log" 2>&1; echo "sweep rc=$?"; tail -4 "$OUT/sweep.log"
ls -la "$OUT/sweep"

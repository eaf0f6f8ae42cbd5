#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_ah; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -k "dense or image or scheme or golden or independent or shard or bf16" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
for rep in 1; do
for mode in auto 128; do
for b in 8 16; do NBASR_ROW_TILE=$mode timeout 300 python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --no-strict 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows', '$mode', 'B', $b, round(d['value']), round(d['ms_per_step'], 3), {k[:6]: (round(v['us_per_launch']), v.get('row_tile')) for k, v in d['roofline_mfma']['per_layer'].items()})"; done
done; done
NBASR_ROW_TILE=auto timeout 300 python bench.py --no-cpu-baseline --no-strict 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B64', round(d['value']), round(d['ms_per_step'], 3), {k[:6]: (round(v['us_per_launch']), v.get('row_tile')) for k, v in d['roofline_mfma']['per_layer'].items()})"

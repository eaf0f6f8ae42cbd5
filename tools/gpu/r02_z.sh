#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_z; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" "$OUT/pytest.log" | tail -2; grep -E "^FAILED" "$OUT/pytest.log" | head
for rep in 1 2; do
for mode in 1 0; do
NBASR_GC_TABLE=$mode timeout 600 python bench.py --no-cpu-baseline --no-strict 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table', $mode, 'B64', round(d['value']), round(d['ms_per_step'], 3), round(d['roofline']['frac'], 4), {k[:6]: round(v['GBps']) for k, v in d['roofline']['per_block'].items()})"
done; done
for mode in 1 0; do
for b in 8 16 32; do NBASR_GC_TABLE=$mode timeout 300 python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --no-strict --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table', $mode, 'B', $b, round(d['value']), round(d['ms_per_step'], 3))"; done
NBASR_GC_TABLE=$mode timeout 300 python bench.py --arch dense-skip --batch 32 --frames 1600 --steps 20 --warmup 5 --no-cpu-baseline --no-strict --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table', $mode, 'dense-skip 32x1600 fp32', round(d['value']), round(d['ms_per_step'], 3))"
NBASR_GC_TABLE=$mode timeout 300 python bench.py --arch dense-skip --batch 32 --frames 1000 --steps 20 --warmup 5 --no-cpu-baseline --no-strict --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table', $mode, 'dense-skip 32x1000 fp32', round(d['value']), round(d['ms_per_step'], 3), d.get('p50_forward_ms'))"
done

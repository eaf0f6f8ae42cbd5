#!/bin/bash
# round 2, GPU call F: flattened vs per-utterance lane mapping of the fp32 node kernel (same box), then the suite
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_f
mkdir -p "$OUT"
export TMPDIR=/tmp
for round in 1 2; do
  NBASR_EXTRA_CXXFLAGS="-DNBASR_GC_FLAT_F32=1" python -m nb_asr_amd.build > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-strict --steps 30 > "$OUT/ab_flat_$round.json" 2>/dev/null
  python -m nb_asr_amd.build > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-strict --steps 30 > "$OUT/ab_zbatch_$round.json" 2>/dev/null
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02_f/ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['value_sequential']), round(d['roofline']['frac'], 4), {k[:6]: round(v['GBps']) for k, v in d['roofline']['per_block'].items()}, round(d['ms_per_forward_by_kernel']['dense_conv'],3))
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 2400 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" "$OUT/pytest.log" | tail -3
grep -E "^FAILED|^ERROR" "$OUT/pytest.log" | head -20

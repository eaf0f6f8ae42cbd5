#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_l; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bf16_ops_gpu.py -x -q -k mfma > "$OUT/pytest_mfma.log" 2>&1; echo "mfma ops rc=$?"; tail -3 "$OUT/pytest_mfma.log"
for sp in 2 3 5 8; do
  NBASR_MFMA_SPLITS=$sp NBASR_GC_BF16_MFMA=1 timeout 600 python bench.py --arch dense-skip --batch 32 --frames 1600 --dtype bf16 --no-cpu-baseline --steps 20 > "$OUT/cfg3_splits$sp.json" 2>/dev/null
  python3 - <<PY
import json
d = json.loads(open('$OUT/cfg3_splits$sp.json').read().strip().splitlines()[-1])
print('splits $sp', round(d['value']), round(d['ms_per_step'],2), round(d['ms_per_forward_by_kernel']['grouped_conv'],2), {k[:6]+k[-16:-13]: round(v['us_per_launch'],1) for k, v in d['roofline']['per_block'].items()})
PY
done

#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_k; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bf16_ops_gpu.py -x -q -k mfma > "$OUT/pytest_mfma.log" 2>&1; echo "mfma ops rc=$?"; tail -3 "$OUT/pytest_mfma.log"
for m in 1 0 1; do
  NBASR_GC_BF16_MFMA=$m timeout 600 python bench.py --arch dense-skip --batch 32 --frames 1600 --dtype bf16 --no-cpu-baseline --steps 30 > "$OUT/cfg3_bf16_mfma$m.json" 2>"$OUT/cfg3_bf16_mfma$m.err"
  python3 - <<PY
import json
try:
    d = json.loads(open('$OUT/cfg3_bf16_mfma$m.json').read().strip().splitlines()[-1])
    print('cfg3 bf16 mfma=$m', round(d['value']), round(d['ms_per_step'],2), 'seq', round(d['value_sequential']), round(d['roofline']['frac'], 4), {k: round(v,2) for k,v in d['ms_per_forward_by_kernel'].items()})
    print('   ', {k[:12]+k[-16:-13]: (round(v['GBps']), round(v['us_per_launch'],1)) for k, v in d['roofline']['per_block'].items()})
except Exception as e:
    print('ERR', e, open('$OUT/cfg3_bf16_mfma$m.err').read()[-800:])
PY
done
done

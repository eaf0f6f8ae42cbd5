#!/bin/bash
# round 2, GPU call E: suite, strict-mode layer noise, store flavour and 2-frame variant A/B
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_e
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" "$OUT/pytest.log" | tail -3
grep -E "^FAILED|^ERROR|exact-fp32 mode|rms err vs fp64" "$OUT/pytest.log" | head -40
NBASR_DENSE_MODE=f32 NBASR_LINEAR_MODE=f32 python tests/layer_noise.py --arch '[[5,1],[3,0,0],[1,0,0,1]]' --batch 3 --frames 130 --no-rnn --seed 502 --xseed 2 > "$OUT/noise_seed2_strict.log" 2>&1; cat "$OUT/noise_seed2_strict.log"
python tests/layer_noise.py --arch '[[5,1],[3,0,0],[1,0,0,1]]' --batch 3 --frames 130 --no-rnn --seed 502 --xseed 2 > "$OUT/noise_seed2_default.log" 2>&1; tail -8 "$OUT/noise_seed2_default.log"
timeout 600 python tools/bench_gc_variants.py --dtypes f32 --json "$OUT/gc_variants_f32.json" > "$OUT/gc_variants_f32.log" 2>&1; grep -E '"block": [23]' "$OUT/gc_variants_f32.log"
for cfg in "base:" "keep100:NBASR_GC_KEEP_MB=100" "keep200:NBASR_GC_KEEP_MB=200" "fpl2:NBASR_GC_F32_VARIANT=4" "base2:" "keep100b:NBASR_GC_KEEP_MB=100"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --no-cpu-baseline --no-strict --steps 30 > "$OUT/ab_$name.json" 2>/dev/null
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02_e/ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['value_sequential']), round(d['roofline']['frac'], 4), {k[:6]: round(v['GBps']) for k, v in d['roofline']['per_block'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY

#!/bin/bash
# round 2, GPU call H: the restored fp32 node kernel in the model, backward building blocks, suite
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_h; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q > "$OUT/pytest_bwd.log" 2>&1; echo "bwd rc=$?"; tail -12 "$OUT/pytest_bwd.log"
timeout 600 python bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r02_h/bench.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['value_sequential']), round(d['value_strict_f32']), round(d['roofline']['frac'], 4), {k[:6]: round(v['GBps']) for k, v in d['roofline']['per_block'].items()})
PY
timeout 900 python tools/ubench/ab_gc_r1.py > "$OUT/ab_gc_r1.log" 2>&1; tail -12 "$OUT/ab_gc_r1.log" | cut -c1-200
timeout 600 python tools/bench_gc_variants.py --batch 32 --frames 1600 --kernel 7 --dilation 2 --dtypes bf16 --json "$OUT/gc_variants_bf16.json" > "$OUT/gc_variants_bf16.log" 2>&1; tail -16 "$OUT/gc_variants_bf16.log" | cut -c1-250
timeout 600 python bench.py --arch dense-skip --batch 32 --frames 1600 --dtype bf16 --no-cpu-baseline > "$OUT/bench_cfg3_bf16.json" 2>/dev/null
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r02_h/bench_cfg3_bf16.json').read().strip().splitlines()[-1])
print('cfg3 bf16', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'], 4), {k: round(v,2) for k,v in d['ms_per_forward_by_kernel'].items()})
PY
timeout 2400 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" "$OUT/pytest.log" | tail -3; grep -E "^FAILED|^ERROR" "$OUT/pytest.log" | head -20

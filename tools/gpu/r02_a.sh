#!/bin/bash
# round 2, GPU call A: full -m gpu suite, bench line, VALU/issue counters of the grouped conv, NaN-propagation A/B
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_a
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
tail -c 600 "$OUT/bench.json"
cd /tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$OUT/valu" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-strict --no-pipeline > "$OUT/valu.log" 2>&1
cd "$REPO"
V=$(find "$OUT/valu" -name '*counter_collection.csv' | head -1)
python3 tools/summarize_valu.py "$V" > "$OUT/r02_pmc_valu_issue.csv" 2> "$OUT/valu_sum.err"
rm -rf "$OUT/valu"
cat "$OUT/r02_pmc_valu_issue.csv" | head -30
# A/B: bare v_med3 clamp (NaN -> 0) vs the NaN-propagating clamp, same box, alternating
for round in 1 2; do
  NBASR_EXTRA_CXXFLAGS="-DNBASR_NAN_QUIET=1" python -m nb_asr_amd.build > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-strict --steps 30 > "$OUT/ab_quiet_$round.json" 2>/dev/null
  python -m nb_asr_amd.build > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-strict --steps 30 > "$OUT/ab_loud_$round.json" 2>/dev/null
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02_a/ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['value_sequential']), round(d['roofline']['frac'], 4), {k: round(v['GBps']) for k, v in d['roofline']['per_block'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY

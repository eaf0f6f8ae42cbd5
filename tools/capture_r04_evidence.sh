#!/bin/bash
# round 4 evidence set: bench lines, rocprofv3 kernel stats, PMC passes, small-batch traces (copy the summaries into profiles/ afterwards)
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r04_final; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/capture_profiles.sh r04 > $OUT/capture.log 2>&1
cp -r $REPO/gpurun_out/profiles_r04/* $OUT/ 2>/dev/null
cd /tmp
for B in 8 16; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_b$B -- python3 $REPO/tools/ubench/small_batch_trace.py --batch $B > $OUT/trace_b$B.log 2>&1
  T=$(find $OUT/trace_b$B -name '*kernel_trace.csv' | head -1)
  python3 $REPO/tools/summarize_gaps.py $T > $OUT/r04_small_batch_trace_b${B}.txt
  tail -2 $OUT/trace_b$B.log
  rm -rf $OUT/trace_b$B
done
cd $REPO
for B in 8 16 32; do
  python3 bench.py --batch $B --no-cpu-baseline --no-strict > $OUT/r04_bench_b$B.json 2> /dev/null
  python3 -c "
import json; d=json.loads(open('$OUT/r04_bench_b$B.json').read().strip().splitlines()[-1]); print($B, round(d['value']), d['ms_per_step'], d.get('p50_forward_ms'))"
done
python3 -c "
import json; d=json.loads(open('$OUT/r04_bench_n1.json').read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), d['ms_per_step'], d['p50_forward_ms'], r['frac'], r['frac_credited_node_ops'], r['traffic'], d['parity']['ok'], d['cpu_baseline']['value'], d['value_strict_f32'])"
python3 -c "
import json; d=json.loads(open('$OUT/r04_bench_cfg3_bf16.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg3', round(d['value']), d['ms_per_step'], r['frac'], d.get('parity'))"
ls $OUT

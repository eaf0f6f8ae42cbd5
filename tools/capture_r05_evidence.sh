#!/bin/bash
# round 5 evidence set on ONE box: bench lines (default, small batches, configs[3] bf16), rocprofv3 kernel stats, PMC passes, the training step,
# and BASELINE configs[4]'s sweep artefact.  Copy the summaries from gpurun_out/r05_final/ into profiles/ afterwards.
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r05_final; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/capture_profiles.sh r05 > $OUT/capture.log 2>&1
cp -r $REPO/gpurun_out/profiles_r05/* $OUT/ 2>/dev/null
cd $REPO
for B in 8 16 32; do
  python3 bench.py --batch $B --no-cpu-baseline --no-strict > $OUT/r05_bench_b$B.json 2> /dev/null
  python3 -c "
import json; d=json.loads(open('$OUT/r05_bench_b$B.json').read().strip().splitlines()[-1]); print($B, round(d['value']), d['ms_per_step'], d.get('p50_forward_ms'))"
done
python3 tools/ubench/train_step.py --batch 64 --frames 1000 --steps 3 > $OUT/r05_train_step.jsonl 2> $OUT/train.err; tail -2 $OUT/r05_train_step.jsonl
python3 -c "
import json; d=json.loads(open('$OUT/r05_bench_n1.json').read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), d['ms_per_step'], d['p50_forward_ms'], d['value_b_over_p50'], r['frac'], r['frac_credited_node_ops'], r['traffic'], d['parity']['ok'], d['parity_strict_f32']['ok'], d['cpu_baseline']['value'], d['value_strict_f32']); print(d['ms_per_forward_by_kernel']); print(d['roofline_mfma'].get('frac'), d['roofline_mfma_strict_f32'].get('frac') if d.get('roofline_mfma_strict_f32') else None)"
python3 -c "
import json; d=json.loads(open('$OUT/r05_bench_cfg3_bf16.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg3', round(d['value']), d['ms_per_step'], r['frac'], d.get('parity'))"
if [ "$1" = "sweep" ]; then
  mkdir -p $OUT/sweep
  python3 tools/latency_sweep.py --out $OUT/sweep --summary $OUT/r05_latency_sweep_summary.json > $OUT/sweep.log 2>&1
  tail -3 $OUT/sweep.log; ls -la $OUT/sweep
fi
ls $OUT

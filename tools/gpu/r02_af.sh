#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_af; mkdir -p "$OUT"; export TMPDIR=/tmp
for rep in 1 2 3; do
for mode in staged direct; do
NBASR_DENSE_EPILOGUE=$mode timeout 600 python bench.py --no-cpu-baseline --no-strict 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', round(d['value']), round(d['ms_per_step'], 3), {k[:6]: round(v['us_per_launch']) for k, v in d['roofline_mfma']['per_layer'].items()}, round(d['ms_per_forward_by_kernel']['dense_conv'], 3))"
done; done

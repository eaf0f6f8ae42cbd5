#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_g; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python tools/ubench/ab_gc_r1.py > "$OUT/ab_gc_r1.log" 2>&1; echo rc=$?; tail -14 "$OUT/ab_gc_r1.log"

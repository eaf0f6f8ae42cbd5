#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_v; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python tools/ubench/gc_phases.py > "$OUT/gc_phases.log" 2>&1; echo rc=$?; tail -6 "$OUT/gc_phases.log"

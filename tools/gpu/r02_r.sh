#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_r; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python tools/ubench/gc_chain.py > "$OUT/gc_chain.log" 2>&1; echo rc=$?; tail -6 "$OUT/gc_chain.log"

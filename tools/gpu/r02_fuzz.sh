#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_fuzz; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 2400 python tests/fuzz_architectures.py 150 5000 > "$OUT/fuzz.log" 2>&1; echo "fuzz rc=$?"; tail -6 "$OUT/fuzz.log"

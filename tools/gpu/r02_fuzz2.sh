#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_fuzz2; mkdir -p "$OUT"; export TMPDIR=/tmp
for s in 5026 5028 5063 5075 5085; do
  echo "== seed $s default"; timeout 300 python tests/fuzz_architectures.py 1 $s 2>&1 | grep -v amdgpu | tail -3
  echo "== seed $s exact fp32 MFMA"; NBASR_DENSE_MODE=f32 NBASR_LINEAR_MODE=f32 timeout 300 python tests/fuzz_architectures.py 1 $s 2>&1 | grep -v amdgpu | tail -3
done > "$OUT/fuzz2.log" 2>&1
cat "$OUT/fuzz2.log"

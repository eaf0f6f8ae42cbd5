#!/bin/bash
# round 2, GPU call C: whole -m gpu suite on the new build, per-layer noise of the case that missed the 1.25x rms rule
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_c
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" "$OUT/pytest.log" | tail -3
grep -E "^FAILED|^ERROR|AssertionError|rms err|worst err/tol vs" "$OUT/pytest.log" | head -60
python tests/layer_noise.py --arch '[[0,1],[5,1,0],[2,0,1,1]]' --batch 2 --frames 258 --no-rnn --seed 77 --xseed 5 > "$OUT/noise_M.log" 2>&1; cat "$OUT/noise_M.log"
NBASR_LINEAR_MODE=f32 python tests/layer_noise.py --arch '[[0,1],[5,1,0],[2,0,1,1]]' --batch 2 --frames 258 --no-rnn --seed 77 --xseed 5 > "$OUT/noise_M_linf32.log" 2>&1; tail -5 "$OUT/noise_M_linf32.log"
python tests/layer_noise.py A_lively_b1_t500 > "$OUT/noise_A.log" 2>&1; tail -8 "$OUT/noise_A.log"

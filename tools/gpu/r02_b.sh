#!/bin/bash
# round 2, GPU call B: bf16 path parity, kernel-variant A/B, bf16 bench of BASELINE configs[3]
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_b
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bf16_ops_gpu.py -x -q > "$OUT/pytest_ops.log" 2>&1; echo "ops rc=$?"; tail -15 "$OUT/pytest_ops.log"
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -s > "$OUT/pytest_model.log" 2>&1; echo "model rc=$?"; grep -E "rms err|worst err|passed|failed|Error|assert" "$OUT/pytest_model.log" | tail -40
timeout 600 python tools/bench_gc_variants.py --json "$OUT/gc_variants_b64_t1000_k5.json" > "$OUT/gc_variants_k5.log" 2>&1; tail -32 "$OUT/gc_variants_k5.log"
timeout 600 python tools/bench_gc_variants.py --batch 32 --frames 1600 --kernel 7 --dilation 2 --dtypes bf16 --json "$OUT/gc_variants_b32_t1600_k7d2.json" > "$OUT/gc_variants_k7.log" 2>&1; tail -16 "$OUT/gc_variants_k7.log"
timeout 600 python bench.py --arch dense-skip --batch 32 --frames 1600 --dtype bf16 --no-cpu-baseline > "$OUT/bench_cfg3_bf16.json" 2> "$OUT/bench_cfg3_bf16.err"; echo "bench bf16 rc=$?"; tail -c 1500 "$OUT/bench_cfg3_bf16.json"; tail -5 "$OUT/bench_cfg3_bf16.err"
timeout 600 python bench.py --arch dense-skip --batch 32 --frames 1600 --no-cpu-baseline --no-strict > "$OUT/bench_cfg3_f32.json" 2> "$OUT/bench_cfg3_f32.err"; echo "bench f32 rc=$?"; tail -c 600 "$OUT/bench_cfg3_f32.json"

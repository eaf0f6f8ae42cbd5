#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_q; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/model" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-strict --no-roofline > "$OUT/model.log" 2>&1; echo rc=$?
f=$(find "$OUT/model" -name '*counter_collection.csv' | head -1)
python3 "$REPO/tools/summarize_mfma.py" "$f" --all > "$OUT/clock_in_model.csv"; grep grouped_conv_f32 "$OUT/clock_in_model.csv"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/iso" -- python3 "$REPO/tools/ubench/ab_gc_r1.py" > "$OUT/iso.log" 2>&1; echo rc=$?
f=$(find "$OUT/iso" -name '*counter_collection.csv' | head -1)
python3 "$REPO/tools/summarize_mfma.py" "$f" --all > "$OUT/clock_isolated.csv"; grep grouped_conv "$OUT/clock_isolated.csv"
rm -rf "$OUT/model" "$OUT/iso"

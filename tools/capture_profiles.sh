#!/bin/bash
# Re-captures the judged evidence under gpurun_out/ (copy the summaries into profiles/ afterwards).
# usage (on the GPU box, from the repo root):  bash tools/capture_profiles.sh r01
# (the profiled runs keep ONE chain of steps in flight -- a second chain's kernels would stretch the durations being recorded -- and leave
# the strong-scaling proxy out: its 8-utterance launches would be averaged into the per-kernel durations and counters)
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/${TAG}_bench_n1.json" 2> "$OUT/bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy --steps 20 --warmup 5 --no-cpu-baseline --no-strict > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy --steps 2 --warmup 1 --no-cpu-baseline --no-strict > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy --steps 2 --warmup 1 --no-cpu-baseline --no-strict > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/mfma" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-strict > "$OUT/mfma.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$OUT/valu" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-strict > "$OUT/valu.log" 2>&1
# BASELINE configs[3] per GPU in bf16: bench line + kernel stats
python3 "$REPO/bench.py" --arch dense-skip --batch 32 --frames 1600 --dtype bf16 > "$OUT/${TAG}_bench_cfg3_bf16.json" 2> "$OUT/bench_bf16.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_bf16" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy --arch dense-skip --batch 32 --frames 1600 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/stats_bf16.log" 2>&1
cd "$REPO"
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
M=$(find "$OUT/mfma" -name '*counter_collection.csv' | head -1)
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
S16=$(find "$OUT/stats_bf16" -name '*kernel_stats.csv' | head -1)
V=$(find "$OUT/valu" -name '*counter_collection.csv' | head -1)
python3 tools/summarize_pmc.py "$F" "$W" > "$OUT/${TAG}_pmc_hbm_traffic.csv"
python3 tools/summarize_mfma.py "$M" > "$OUT/${TAG}_pmc_mfma_utilisation.csv" 2> "$OUT/mfma_sum.err"
cp "$OUT/${TAG}_pmc_hbm_traffic.csv" "$REPO/profiles/${TAG}_pmc_hbm_traffic.csv"      # bench.py reads the newest committed summary
python3 "$REPO/bench.py" > "$OUT/${TAG}_bench_n1.json" 2> "$OUT/bench.err"               # the line with traffic from this capture
cp "$S" "$OUT/${TAG}_kernel_stats_full.csv"
cp "$S16" "$OUT/${TAG}_kernel_stats_cfg3_bf16_full.csv"
python3 tools/summarize_valu.py "$V" > "$OUT/${TAG}_pmc_valu_issue.csv" 2> "$OUT/valu_sum.err"
python3 tools/summarize_kernels.py "$S" > "$OUT/${TAG}_kernel_stats.csv" 2>> "$OUT/valu_sum.err"
python3 tools/summarize_kernels.py "$S16" > "$OUT/${TAG}_kernel_stats_cfg3_bf16.csv" 2>> "$OUT/valu_sum.err"
# keep the traces themselves out of the merged gpurun_out (size)
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/mfma" "$OUT/valu" "$OUT/stats_bf16"
ls -la "$OUT"

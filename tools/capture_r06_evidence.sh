#!/bin/bash
# Round 6 evidence set on ONE box.  Copy the summaries from gpurun_out/r06_final/ into profiles/ afterwards.
#   bench lines: default (with cpu_baseline, strong proxy), --force-collective, 8 / 16 / 32 utterances, configs[3] bf16
#   rocprofv3: kernel stats (default, bf16), PMC passes (HBM traffic, matrix pipe, vector issue, LDS) -- counters in their own runs
#   micro-benchmarks: the three recurrence forms, the resident form's phase stamps, what the pipelined tail costs the encoder
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06_final; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/capture_profiles.sh r06 > $OUT/capture.log 2>&1
cp -r $REPO/gpurun_out/profiles_r06/* $OUT/ 2>/dev/null
BF16="--arch dense-skip --batch 32 --frames 1600 --dtype bf16"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch16" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy $BF16 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/fetch16.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write16" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy $BF16 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/write16.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/mfma16" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy $BF16 --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline > "$OUT/mfma16.log" 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$OUT/lds16" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy $BF16 --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline > "$OUT/lds16.log" 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$OUT/lds32" -- python3 "$REPO/bench.py" --in-flight 1 --no-strong-proxy --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-strict > "$OUT/lds32.log" 2>&1
cd $REPO
F=$(find "$OUT/fetch16" -name '*counter_collection.csv' | head -1); W=$(find "$OUT/write16" -name '*counter_collection.csv' | head -1)
python3 tools/summarize_pmc.py "$F" "$W" > "$OUT/r06_pmc_hbm_traffic_cfg3_bf16.csv"
python3 tools/summarize_mfma.py "$(find "$OUT/mfma16" -name '*counter_collection.csv' | head -1)" > "$OUT/r06_pmc_mfma_utilisation_cfg3_bf16.csv"
python3 tools/summarize_lds.py "$(find "$OUT/lds16" -name '*counter_collection.csv' | head -1)" > "$OUT/r06_pmc_lds_cfg3_bf16.csv"
python3 tools/summarize_lds.py "$(find "$OUT/lds32" -name '*counter_collection.csv' | head -1)" > "$OUT/r06_pmc_lds.csv"
cp "$OUT/r06_pmc_hbm_traffic_cfg3_bf16.csv" profiles/r06_pmc_hbm_traffic_cfg3_bf16.csv
python3 bench.py $BF16 > $OUT/r06_bench_cfg3_bf16.json 2> $OUT/bench_bf16.err          # the bf16 line with traffic from this capture
rm -rf "$OUT/fetch16" "$OUT/write16" "$OUT/mfma16" "$OUT/lds16" "$OUT/lds32"
for B in 8 16 32; do
  python3 bench.py --batch $B --no-cpu-baseline --no-strict > $OUT/r06_bench_b$B.json 2> /dev/null
done
MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python3 bench.py --force-collective --no-cpu-baseline --no-strict --no-roofline > $OUT/r06_bench_force_collective.json 2> $OUT/fc.err
python3 tools/ubench/lstm_xcd.py 8 16 32 64 128 > $OUT/r06_lstm_forms.txt 2>&1
python3 tools/ubench/lstm_xcd_stamps.py 64 > $OUT/r06_lstm_xcd_stamps.txt 2>&1
python3 tools/ubench/lstm_xcd_stamps.py 8 >> $OUT/r06_lstm_xcd_stamps.txt 2>&1
python3 tools/ubench/cell_os.py --batches 4 8 16 64 > $OUT/r06_cell_output_channel_split.txt 2>&1
for B in 8 16 64; do echo "== batch $B, two whole batches in flight on two streams (split) against one chain (whole)"; python3 tools/ubench/two_half_batches.py --batch $B --full --ways 2 --steps 50 2>&1 | grep -v amdgpu; done > $OUT/r06_two_chains_in_flight.txt
python3 tools/ubench/in_flight_check.py --overlap-main --steps 60 --rounds 3 2>&1 | grep -v amdgpu > $OUT/r06_in_flight_check.txt
python3 tools/ubench/tail_cost.py > $OUT/r06_tail_cost.txt 2>&1
python3 tools/ubench/tail_cost.py --skip >> $OUT/r06_tail_cost.txt 2>&1
python3 tools/ubench/train_step.py --batch 64 --frames 1000 --steps 3 > $OUT/r06_train_step.jsonl 2> $OUT/train.err
python3 -c "
import json
for f in ('r06_bench_n1', 'r06_bench_cfg3_bf16', 'r06_bench_b8', 'r06_bench_b16', 'r06_bench_b32', 'r06_bench_force_collective'):
    d = json.loads(open('$OUT/' + f + '.json').read().strip().splitlines()[-1]); r = d.get('roofline') or {}
    print(f, round(d['value']), 'one chain', round(d['value_one_in_flight']), round(d['ms_per_step'], 3), round(d['p50_forward_ms'], 3), r.get('frac'), r.get('traffic'), (d.get('roofline_mfma') or {}).get('frac'), d.get('strong_proxy') and (round(d['strong_proxy']['value']), round(d['strong_proxy']['projected_x8'], 2)), d.get('allgather_us'), (d.get('parity') or {}).get('ok'))
"
grep -h "full\|skip" $OUT/r06_tail_cost.txt | tail -6
ls $OUT

# BASELINE configs[4] on this build (one GPU) and the parity sweep
if [ "$1" = "sweep" ]; then
  python3 tools/latency_sweep.py --out $OUT/r06_sweep --summary $OUT/r06_latency_sweep_summary.json > $OUT/sweep.log 2>&1; tail -2 $OUT/sweep.log
  timeout 1500 python3 tests/fuzz_architectures.py 150 5000 > $OUT/r06_fuzz_150_seeds_5000.txt 2> $OUT/fuzz.err; tail -1 $OUT/r06_fuzz_150_seeds_5000.txt
fi

#!/bin/bash
# round 4, after the node-kernel fix: the PMC traffic passes, the headline line and smoke() on the final build
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r04_refresh; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-strict > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-strict > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-strict > "$OUT/stats.log" 2>&1
cd $REPO
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
python3 tools/summarize_pmc.py "$F" "$W" > "$OUT/r04_pmc_hbm_traffic.csv"
python3 tools/summarize_kernels.py "$S" > "$OUT/r04_kernel_stats.csv"
cp "$OUT/r04_pmc_hbm_traffic.csv" profiles/r04_pmc_hbm_traffic.csv
python3 bench.py > "$OUT/r04_bench_n1.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$OUT/r04_bench_n1.json').read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), d['ms_per_step'], r['frac'], r['frac_credited_node_ops'], r['traffic'], d['parity']['ok'], d['cpu_baseline']['value'], d.get('build_id'))"
python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/stats"

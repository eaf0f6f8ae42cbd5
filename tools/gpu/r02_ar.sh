#!/bin/bash
# End-of-round confirmation on the final tree: full GPU suite, smoke(), the driver's two bench invocations.
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_ar; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" "$OUT/pytest.log" | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300

#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_m; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python tools/ubench/host_issue.py > "$OUT/host_issue.log" 2>&1; echo rc=$?; tail -8 "$OUT/host_issue.log"

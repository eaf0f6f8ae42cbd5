#!/bin/bash
# round 4: the 150-seed random sweep of round 2 / 3 on the final build (copy the record into profiles/ afterwards)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_fuzz; mkdir -p $OUT
S=$(date +%s)
timeout 1100 python3 tests/fuzz_architectures.py 150 5000 > $OUT/fuzz_150_seeds_5000.txt 2> $OUT/fuzz.err; echo "rc=$? $(( $(date +%s) - S )) s"
tail -1 $OUT/fuzz_150_seeds_5000.txt
python3 -c "import nb_asr_amd.build as b; print('build', b.source_hash())"

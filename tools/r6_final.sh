#!/bin/bash
# round 6 closing run: AHDS step profile, the default bench line, the full GPU suite
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6z; mkdir -p $OUT
bash tools/prof_ahds.sh > $OUT/prof_ahds.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
(time timeout 2400 python -m pytest tests -q -m gpu) > $OUT/gpu_tests.txt 2>&1
tail -4 $OUT/gpu_tests.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6b; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_sh_mfma.py tests/test_gpu_raster_parity.py -x -q -m gpu 2>&1 | tail -15 > $OUT/tests.txt
cat $OUT/tests.txt
timeout 600 python tools/exp_sh_mfma.py > $OUT/sh_mfma.txt 2>$OUT/sh_mfma.err
cat $OUT/sh_mfma.txt; tail -3 $OUT/sh_mfma.err

#!/bin/bash
python tools/bench_refine.py > gpurun_out/r4_refine.txt 2>&1
bash tools/ab_ahds.sh "GIP_SPLITK_STATS=0" "GIP_SPLITK_STATS=1" > gpurun_out/r4_ab_splitk_stats.txt 2>&1

#!/bin/bash
# round 5, GPU run 13: Winograd GN test again; kernel profile of ONE rank's shard of a 4-rank configs[3] group (1 view: denoise at batch 3, VAE at batch 1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "winograd" > gpurun_out/r5/run13_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5/run13_tests.log
tail -3 gpurun_out/r5/run13_tests.log
python tools/bench_ahds.py --steps 10 --warmup 4 --proxy-group 4 2>/dev/null | tee gpurun_out/r5/run13_proxy4.json | cut -c1-600
bash tools/prof_ahds.sh --proxy-group 4
cd $GRAFT_REPO_ROOT
cp gpurun_out/profiles_new/ahds_step_summary.txt gpurun_out/r5/run13_shard1_summary.txt
cp gpurun_out/profiles_new/ahds_step_trace.txt gpurun_out/r5/run13_shard1_trace.txt

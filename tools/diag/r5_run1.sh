#!/bin/bash
# round 5, GPU run 1: validate the hygiene changes + new tests, take the new measurement objects
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_attention.py tests/test_gpu_glue.py tests/test_gpu_groupnorm.py tests/test_gpu_network_parity.py tests/test_gpu_anpg_sensitivity.py -x -q -m gpu -s > gpurun_out/r5/run1_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5/run1_tests.log
tail -5 gpurun_out/r5/run1_tests.log
python -m pytest tests/test_gpu_kernel_whitelist.py -q -m gpu -s > gpurun_out/r5/run1_whitelist.log 2>&1
tail -3 gpurun_out/r5/run1_whitelist.log
python bench.py > gpurun_out/r5/run1_bench.json 2> gpurun_out/r5/run1_bench.err
echo "bench rc $?"
cut -c1-400 gpurun_out/r5/run1_bench.json

#!/bin/bash
python -m pytest tests/test_gpu_conv.py tests/test_gpu_groupnorm.py tests/test_gpu_ahds_step.py tests/test_gpu_network_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4_run48_tests.txt
python tools/exp_vae_time.py 2>&1 | grep -v amdgpu | tail -6 >> gpurun_out/r4_run48_tests.txt

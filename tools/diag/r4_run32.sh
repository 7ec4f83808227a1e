#!/bin/bash
python -m pytest tests/test_gpu_glue.py tests/test_gpu_pipeline.py tests/test_gpu_ahds_step.py tests/test_gpu_sharded_step.py tests/test_gpu_network_parity.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4_run32_tests.txt
bash tools/ab_ahds.sh "GIP_FUSED_GLUE=0 GIP_FUSED_LOSS=0 GIP_FUSED_ACTIVATIONS=0 GIP_POSE_STREAM=0" "GIP_X=1" "GIP_FUSED_GLUE=0 GIP_FUSED_LOSS=0 GIP_FUSED_ACTIVATIONS=0 GIP_POSE_STREAM=0" "GIP_X=1" > gpurun_out/r4_ab_glue3.txt 2>&1
bash tools/prof_ahds.sh > gpurun_out/r4_prof_ahds3.log 2>&1

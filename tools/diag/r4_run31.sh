#!/bin/bash
python -m pytest tests/test_gpu_glue.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r4_run31_tests.txt
bash tools/ab_ahds.sh "GIP_FUSED_GLUE=0 GIP_FUSED_LOSS=0 GIP_FUSED_ACTIVATIONS=0 GIP_POSE_STREAM=0" "GIP_X=1" "GIP_FUSED_GLUE=0 GIP_FUSED_LOSS=0 GIP_FUSED_ACTIVATIONS=0 GIP_POSE_STREAM=0" "GIP_X=1" "GIP_FUSED_GLUE=0 GIP_FUSED_LOSS=0 GIP_FUSED_ACTIVATIONS=0 GIP_POSE_STREAM=0" "GIP_X=1" > gpurun_out/r4_ab_glue2.txt 2>&1
bash tools/prof_ahds.sh > gpurun_out/r4_prof_ahds2.log 2>&1

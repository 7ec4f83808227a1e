#!/bin/bash
# glue kernels + pose stream + 8-wave attention: tests, then same-box A/Bs of the AHDS step
python -m pytest tests/test_gpu_glue.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4_glue_tests.txt
for x in 0 1 0 1; do echo "GIP_ATTN_XCD=$x"; GIP_ATTN_XCD=$x python tools/exp_attn_split.py 2>&1 | grep -v "amdgpu\|SPLIT"; done > gpurun_out/r4_attn_xcd.txt 2>&1
bash tools/ab_ahds.sh "GIP_FUSED_GLUE=0 GIP_POSE_STREAM=0 GIP_ATTN_NW=4" "GIP_X=1" "GIP_FUSED_GLUE=0" "GIP_POSE_STREAM=0" "GIP_ATTN_NW=4" "GIP_FUSED_GLUE=0 GIP_POSE_STREAM=0 GIP_ATTN_NW=4" "GIP_X=1" > gpurun_out/r4_ab_glue.txt 2>&1
python -m pytest tests/test_gpu_attention.py tests/test_gpu_ahds_step.py tests/test_gpu_network_parity.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/r4_glue_tests.txt

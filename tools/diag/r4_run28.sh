#!/bin/bash
python -m pytest tests/test_gpu_glue.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4_glue_tests2.txt
bash tools/ab_ahds.sh "GIP_FUSED_LOSS=0 GIP_FUSED_ACTIVATIONS=0" "GIP_X=1" "GIP_FUSED_LOSS=0 GIP_FUSED_ACTIVATIONS=0" "GIP_X=1" > gpurun_out/r4_ab_loss.txt 2>&1
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 >> gpurun_out/r4_glue_tests2.txt

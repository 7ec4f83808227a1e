#!/bin/bash
# q2 attention adopted: attention + network parity + refine tests, then the AHDS A/B (GIP_ATTN_QT=1 = previous dispatch)
python -m pytest tests/test_gpu_attention.py tests/test_gpu_network_parity.py tests/test_gpu_refine.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r4_run40_tests.txt
bash tools/ab_ahds.sh "GIP_ATTN_QT=1" "GIP_X=1" "GIP_ATTN_QT=1" "GIP_X=1" > gpurun_out/r4_ab_q2.txt 2>&1

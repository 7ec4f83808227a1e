#!/bin/bash
python -m pytest tests/test_gpu_attention.py tests/test_gpu_network_parity.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4_run50_tests.txt
GIP_ATTN_REF=1 GIP_ATTN_NW=4 GIP_ATTN_QT=1 python tools/diag/attn_variant_check.py 2>&1 | grep -v amdgpu | head -4 >> gpurun_out/r4_run50_tests.txt
python tools/diag/attn_variant_check.py 2>&1 | grep -v amdgpu | head -4 >> gpurun_out/r4_run50_tests.txt

#!/bin/bash
# two query tiles per wave (GIP_ATTN_QT=2) at 4 / 8 waves per workgroup against the shipped dispatch
GIP_ATTN_REF=1 GIP_ATTN_NW=4 python tools/diag/attn_variant_check.py 2>&1 | grep -v amdgpu > gpurun_out/r4_attn_q2.txt
for rep in 1 2; do
for cfg in "GIP_X=1" "GIP_ATTN_QT=2 GIP_ATTN_NW=4" "GIP_ATTN_QT=2 GIP_ATTN_NW=8"; do env $cfg python tools/diag/attn_variant_check.py 2>&1 | grep -v amdgpu; done
done >> gpurun_out/r4_attn_q2.txt 2>&1

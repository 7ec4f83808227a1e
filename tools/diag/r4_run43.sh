#!/bin/bash
GIP_ATTN_REF=1 python tools/diag/attn_cross_check.py 2>&1 | grep -v amdgpu > gpurun_out/r4_attn_cross.txt
for rep in 1 2; do for c in 0 1; do GIP_ATTN_CROSS8=$c python tools/diag/attn_cross_check.py 2>&1 | grep -v amdgpu; done; done >> gpurun_out/r4_attn_cross.txt 2>&1

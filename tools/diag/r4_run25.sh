#!/bin/bash
# what bounds the D = 40 attention loop: builds with parts removed (wrong results, timing only): 1 = no exp2, 2 = no PV MFMAs, 4 = no S^T MFMAs
for lib in libgip_nn_base.so libgip_nn_d1.so libgip_nn_d2.so libgip_nn_d4.so libgip_nn_d6.so libgip_nn_d7.so; do echo $lib; GIP_NN_LIB=$lib python tools/exp_attn_split.py 2>&1 | grep -v "amdgpu\|SPLIT" | head -3; done > gpurun_out/r4_attn_diag.txt 2>&1

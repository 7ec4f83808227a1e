#!/bin/bash
# same-box A/B of three small changes to the attention loop (AT_OPT bits: 1 = half-wave maximum by v_permlane32_swap instead of
# ds_bpermute, 2 = the two S^T accumulation chains interleaved, 4 = s_setprio 1 around the S^T MFMAs)
for lib in libgip_nn_base.so libgip_nn_o1.so libgip_nn_o3.so libgip_nn_o7.so libgip_nn_base.so libgip_nn_o1.so libgip_nn_o3.so libgip_nn_o7.so; do echo $lib; GIP_NN_LIB=$lib python tools/exp_attn_split.py 2>&1 | grep -v "amdgpu\|SPLIT"; done > gpurun_out/r4_attn_opt.txt 2>&1
GIP_NN_LIB=libgip_nn_o7.so python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/r4_attn_opt.txt

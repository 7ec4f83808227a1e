#!/bin/bash
# same-box A/B: exponent arguments of the attention softmax by v_pk_fma_f32 (AT_PK_FMA=1) against 32 scalar v_fma_f32
for lib in libgip_nn_base.so libgip_nn_pk.so libgip_nn_base.so libgip_nn_pk.so; do echo $lib; GIP_NN_LIB=$lib python tools/exp_attn_split.py 2>&1 | grep -v amdgpu; done > gpurun_out/r4_attn_pk.txt 2>&1
GIP_NN_LIB=libgip_nn_pk.so python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/r4_attn_pk.txt

#!/bin/bash
# attention with NW waves of 32 queries per workgroup sharing the K / V stages (the stream from L2 per query ~ 1 / NW)
for rep in 1 2; do
for cfg in "4 libgip_nn_base.so" "6 libgip_nn_base.so" "8 libgip_nn_base.so" "8 libgip_nn_w4.so"; do set -- $cfg; echo "GIP_ATTN_NW=$1 $2"; GIP_ATTN_NW=$1 GIP_NN_LIB=$2 python tools/exp_attn_split.py 2>&1 | grep -v "amdgpu\|SPLIT"; done
done > gpurun_out/r4_attn_nw.txt 2>&1
for nw in 6 8; do GIP_ATTN_NW=$nw python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/r4_attn_nw.txt; done

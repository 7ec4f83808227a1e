python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r4_run14_tests.log
for i in 1 2; do
GIP_ATTN_SPLIT=1 python tools/exp_attn_split.py >> gpurun_out/r4_attn_split.txt 2>&1
GIP_ATTN_SPLIT=0 python tools/exp_attn_split.py >> gpurun_out/r4_attn_split.txt 2>&1
done

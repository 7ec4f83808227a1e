for i in 1 2; do
GIP_NN_LIB=libgip_nn_attn4.so python tools/exp_attn_split.py 2>&1 | grep -v amdgpu | sed 's/^/attn4 /' >> gpurun_out/r4_attn4.txt
python tools/exp_attn_split.py 2>&1 | grep -v amdgpu | sed 's/^/base  /' >> gpurun_out/r4_attn4.txt
done

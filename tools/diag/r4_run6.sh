python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4_run6_tests.log
python tools/exp_conv_halo.py > gpurun_out/r4_halo_on.txt 2>&1
GIP_CONV_HALO=0 python tools/exp_conv_halo.py > gpurun_out/r4_halo_off.txt 2>&1
for i in 1 2; do
python tools/exp_vae_time.py >> gpurun_out/r4_vae_ab.txt 2>&1
GIP_CONV_HALO=0 python tools/exp_vae_time.py >> gpurun_out/r4_vae_ab.txt 2>&1
done

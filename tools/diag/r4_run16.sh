python -m pytest tests/test_gpu_conv.py tests/test_gpu_groupnorm.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4_run16_tests.log
for m in 1 2 0; do GIP_CONV_HALO=$m python tools/exp_conv_halo.py >> gpurun_out/r4_halo_big.txt 2>&1; done
for i in 1 2; do for m in 1 2 0; do GIP_CONV_HALO=$m python tools/exp_vae_time.py >> gpurun_out/r4_vae_ab2.txt 2>&1; done; done
python -m pytest tests/test_gpu_network_parity.py -x -q -m gpu -k encode 2>&1 | tail -3 >> gpurun_out/r4_run16_tests.log

python -m pytest tests/test_gpu_groupnorm.py tests/test_gpu_conv.py tests/test_gpu_attention.py tests/test_gpu_network_parity.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4_run11_tests.log
python tools/diag/gn_bandwidth.py > gpurun_out/r4_gn_bandwidth3.txt 2>&1
for i in 1 2; do
python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_rcp.txt
GIP_NN_LIB=libgip_nn_base.so python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_rcp.txt
done
python -m pytest tests/test_gpu_sharded_step.py -x -q -m gpu -k real_guidance 2>&1 | tail -3 >> gpurun_out/r4_run11_tests.log

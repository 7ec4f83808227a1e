python -m pytest tests/test_gpu_pipeline.py -x -q -s -m gpu -k "knn" 2>&1 | tail -8 > gpurun_out/r4_run8_tests.log
python -m pytest tests/test_gpu_conv.py tests/test_gpu_attention.py tests/test_gpu_network_parity.py tests/test_gpu_ahds_step.py -x -q -m gpu 2>&1 | tail -8 >> gpurun_out/r4_run8_tests.log
for i in 1 2; do
GIP_OWN_GEMM=1 python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('own=1', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_gemm.txt
GIP_OWN_GEMM=0 python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('own=0', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_gemm.txt
done

python -m pytest tests/test_gpu_ln_fold.py -x -q -s -m gpu 2>&1 | grep -v Warn | tail -25 > gpurun_out/r4_run13_tests.log
python -m pytest tests/test_gpu_network_parity.py tests/test_gpu_attention.py tests/test_gpu_ahds_step.py -x -q -m gpu 2>&1 | tail -6 >> gpurun_out/r4_run13_tests.log
for i in 1 2; do
GIP_LN_FOLD=1 python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lnfold=1', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_lnfold.txt
GIP_LN_FOLD=0 python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lnfold=0', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_lnfold.txt
done

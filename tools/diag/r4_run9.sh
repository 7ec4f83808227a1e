python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "halo" 2>&1 | tail -15 > gpurun_out/r4_run9_tests.log
for i in 1 2; do
GIP_GEGLU_MIN_ROWS=12288 python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('geglu>=12288', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_geglu.txt
python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_geglu.txt
done
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4_run9_full.log

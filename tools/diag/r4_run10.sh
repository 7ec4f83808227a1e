python -m pytest tests/test_gpu_sharded_step.py -x -q -m gpu -k real_guidance 2>&1 | grep -v Warn | tail -30 | cut -c1-3000 > gpurun_out/r4_run10_sharded.log
python -m pytest tests/test_gpu_raster_parity.py tests/test_gpu_groupnorm.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4_run10_tests.log
python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_config4.py -x -q -m gpu 2>&1 | tail -6 >> gpurun_out/r4_run10_tests.log
for i in 1 2 3; do
GIP_FWD_XCD_ORDER=1 python bench.py --no-ahds --no-cpu-baseline --no-trained --no-exact --repeats 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcd=1', d['ms_per_step'], d['roofline']['stage_ms'])" >> gpurun_out/r4_ab_xcd.txt
GIP_FWD_XCD_ORDER=0 python bench.py --no-ahds --no-cpu-baseline --no-trained --no-exact --repeats 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcd=0', d['ms_per_step'], d['roofline']['stage_ms'])" >> gpurun_out/r4_ab_xcd.txt
done
for i in 1 2; do
GIP_GN_FOLD_FINALIZE=1 python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=1', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_fold.txt
GIP_GN_FOLD_FINALIZE=0 python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=0', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_fold.txt
done

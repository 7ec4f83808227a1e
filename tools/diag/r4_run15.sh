python bench.py > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err
for i in 1 2; do
GIP_GEGLU_MIN_ROWS=12288 python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('geglu>=12288', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_geglu2.txt
python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['denoise_ms'], d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r4_ab_geglu2.txt
done
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4_run15_full.log

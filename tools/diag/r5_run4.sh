#!/bin/bash
# round 5, GPU run 4: network parity after the attention-scale fix, GN-in with register-resident scale / shift, scan without the system fence
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_network_parity.py tests/test_gpu_conv.py tests/test_gpu_raster_parity.py tests/test_gpu_scale.py -x -q -m gpu -s > gpurun_out/r5/run4_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5/run4_tests.log
tail -4 gpurun_out/r5/run4_tests.log
for rep in 1 2 3; do
for cfg in "GIP_CONV_GNIN=0" "GIP_CONV_GNIN=1"; do
  env $cfg python tools/exp_vae_time.py 2>/dev/null | tail -1 | sed "s/^/$cfg /" >> gpurun_out/r5/run4_ab_vae.txt
done
done
cat gpurun_out/r5/run4_ab_vae.txt
for rep in 1 2; do
for lib in libgip_raster_r4.so libgip_raster.so; do
  GIP_RASTER_LIB=$lib python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-ahds --no-trained --no-exact --no-config4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib', d['ms_per_step'], d['roofline']['stage_ms_instrumented'])" >> gpurun_out/r5/run4_ab.txt
done
done
cat gpurun_out/r5/run4_ab.txt
python tools/exp_tile_hist.py > gpurun_out/r5/run4_tile_hist.txt 2>&1; cat gpurun_out/r5/run4_tile_hist.txt | tail -12

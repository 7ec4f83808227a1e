#!/bin/bash
# round 5, GPU run 3: GN-in halo convolution, batched own GEMM, whitelist / strict step; same-box A/Bs of the AHDS step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_conv.py tests/test_gpu_attention.py tests/test_gpu_kernel_whitelist.py tests/test_gpu_network_parity.py tests/test_gpu_ahds_step.py tests/test_gpu_groupnorm.py -x -q -m gpu > gpurun_out/r5/run3_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5/run3_tests.log
tail -8 gpurun_out/r5/run3_tests.log
python tools/exp_winograd_gemm.py > gpurun_out/r5/run3_winograd_gemm.txt 2>&1
tail -25 gpurun_out/r5/run3_winograd_gemm.txt
for rep in 1 2; do
for cfg in "GIP_CONV_GNIN=0" "GIP_CONV_GNIN=1" "GIP_CONV_GNIN=1 GIP_WINOGRAD_GEMM=own"; do
  env $cfg python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$cfg', d['ms_per_step'], 'denoise', d['denoise_ms'], 'vae', d['vae_enc_fwd_bwd_ms'])" >> gpurun_out/r5/run3_ab_ahds.txt
done
done
cat gpurun_out/r5/run3_ab_ahds.txt
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-ahds --no-trained --no-exact --no-config4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('raster', d['ms_per_step'], d['roofline']['stage_ms_instrumented'])"

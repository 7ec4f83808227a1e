#!/bin/bash
# round 5, GPU run 8: after reverting the spilling refactor — pipelined GroupNorm backward vs the same build without, GN-in on / off, AHDS
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_groupnorm.py tests/test_gpu_conv.py tests/test_gpu_network_parity.py -x -q -m gpu > gpurun_out/r5/run8_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5/run8_tests.log
tail -4 gpurun_out/r5/run8_tests.log
for rep in 1 2 3; do
for cfg in "GIP_CONV_GNIN=0" "GIP_CONV_GNIN=1" "GIP_NN_LIB=libgip_nn_nopipe.so"; do
  env $cfg python tools/exp_vae_time.py 2>/dev/null | tail -1 | sed "s/^/$cfg /" >> gpurun_out/r5/run8_ab_vae.txt
done
done
cat gpurun_out/r5/run8_ab_vae.txt
for rep in 1 2; do
  python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('ahds', d['ms_per_step'], 'denoise', d['denoise_ms'], 'vae', d['vae_enc_fwd_bwd_ms'])"
done

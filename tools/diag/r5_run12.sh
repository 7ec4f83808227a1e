#!/bin/bash
# round 5, GPU run 12: GroupNorm inside the Winograd input transform — tests, then same-box A/B of the AHDS step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_conv.py tests/test_gpu_network_parity.py tests/test_gpu_guidance_math.py tests/test_gpu_kernel_whitelist.py tests/test_gpu_ahds_step.py -x -q -m gpu > gpurun_out/r5/run12_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5/run12_tests.log
tail -6 gpurun_out/r5/run12_tests.log
for rep in 1 2 3; do
for cfg in "GIP_WINOGRAD_GN=0" "GIP_WINOGRAD_GN=1"; do
  env $cfg python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$cfg', d['ms_per_step'], 'denoise', d['denoise_ms'], 'vae', d['vae_enc_fwd_bwd_ms'])"
done
done

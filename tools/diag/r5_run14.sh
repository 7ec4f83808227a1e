#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python tools/exp_conv_deep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/run14_conv_deep.txt
for rep in 1 2; do
for cfg in "GIP_CONV_DEEP=0" "GIP_CONV_DEEP=256"; do
  env $cfg python tools/bench_ahds.py --steps 10 --warmup 4 --proxy-group 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('shard1 $cfg', d['ms_per_step'])"
done
done
for cfg in "GIP_CONV_DEEP=0" "GIP_CONV_DEEP=256"; do
  env $cfg python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('full $cfg', d['ms_per_step'], 'denoise', d['denoise_ms'], 'vae', d['vae_enc_fwd_bwd_ms'])"
done

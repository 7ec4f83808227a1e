#!/bin/bash
# same-box A/B of the AHDS step under environment switches: tools/ab_ahds.sh "VAR=val ..." "VAR=val ..."
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$cfg', d['ms_per_step'], 'denoise', d['denoise_ms'], 'vae', d['vae_enc_fwd_bwd_ms'])"
  done
done

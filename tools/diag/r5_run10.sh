#!/bin/bash
# round 5, GPU run 10: does PyTorch TunableOp find faster hipBLASLt / rocBLAS solutions for the step's library GEMMs?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
run() { env $1 python tools/bench_ahds.py --steps 10 --warmup 4 2>gpurun_out/r5/run10_err_$2.txt | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$2', d['ms_per_step'], 'denoise', d['denoise_ms'], 'vae', d['vae_enc_fwd_bwd_ms'])"; }
run "GIP_X=0" default
run "GIP_TUNABLEOP=gpurun_out/r5/tunableop.csv GIP_TUNABLEOP_TUNE=1" tuning
run "GIP_TUNABLEOP=gpurun_out/r5/tunableop.csv" tuned
run "GIP_X=0" default
run "GIP_TUNABLEOP=gpurun_out/r5/tunableop.csv" tuned
ls -la gpurun_out/r5/tunableop*; head -5 gpurun_out/r5/tunableop.csv; wc -l gpurun_out/r5/tunableop.csv

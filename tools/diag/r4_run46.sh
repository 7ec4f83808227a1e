#!/bin/bash
# VAE encoder on two streams (GIP_VAE_STREAMS=2): parity tests under the switch, then same-box A/B
GIP_VAE_STREAMS=2 python -m pytest tests/test_gpu_network_parity.py tests/test_gpu_glue.py -x -q -m gpu -k "encode or vae or guidance_call" 2>&1 | tail -5 > gpurun_out/r4_run46_tests.txt
bash tools/ab_ahds.sh "GIP_VAE_STREAMS=1" "GIP_VAE_STREAMS=2" "GIP_VAE_STREAMS=1" "GIP_VAE_STREAMS=2" > gpurun_out/r4_ab_vae_streams.txt 2>&1

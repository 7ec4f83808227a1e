#!/bin/bash
# GroupNorm statistics folded into the apply kernel's prologue (coalesced per-channel sums): correctness, then same-box A/B
GIP_GN_FOLD_FINALIZE=1 python -m pytest tests/test_gpu_network_parity.py tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4_run45_tests.txt
bash tools/ab_ahds.sh "GIP_GN_FOLD_FINALIZE=0" "GIP_GN_FOLD_FINALIZE=1" "GIP_GN_FOLD_FINALIZE=0" "GIP_GN_FOLD_FINALIZE=1" > gpurun_out/r4_ab_gnfold.txt 2>&1

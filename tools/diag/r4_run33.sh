#!/bin/bash
python -m pytest tests/test_gpu_glue.py -x -q -m gpu -s -k "guidance_call or timestep" 2>&1 | grep -v "Warning\|warn" | tail -40 > gpurun_out/r4_run33_tests.txt

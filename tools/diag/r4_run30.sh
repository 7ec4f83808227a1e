#!/bin/bash
# Winograd GEMMs library vs own kernel; glue tests on the final tree; default bench line; profiled AHDS step summary
python tools/exp_winograd_gemm.py 2>&1 | grep -v amdgpu > gpurun_out/r4_winograd_gemm.txt
python -m pytest tests/test_gpu_glue.py tests/test_gpu_ahds_step.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r4_run30_tests.txt
python bench.py > gpurun_out/r4_bench_final2.json 2> gpurun_out/r4_bench_final2.err
bash tools/prof_ahds.sh > gpurun_out/r4_prof_ahds.log 2>&1

python tools/exp_gemm_table.py > gpurun_out/r4_gemm_table.txt 2>&1
python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4_run7_tests.log

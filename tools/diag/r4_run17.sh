python bench.py --gpus 2 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline --no-trained --no-exact --ahds-steps 4 > gpurun_out/r4_bench_n2.json 2> gpurun_out/r4_bench_n2.err
echo "rc=$?" >> gpurun_out/r4_bench_n2.err
python -m pytest tests/test_gpu_conv.py tests/test_gpu_refine.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r4_run17_tests.log

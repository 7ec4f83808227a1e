python -m pytest tests/test_gpu_sharded_step.py -x -q -s -m gpu 2>&1 | tail -30 > gpurun_out/r4_sharded.log
GIP_TORCH_PROFILE=gpurun_out/r4_ahds_ops.txt python tools/bench_ahds.py --steps 6 --warmup 4 > gpurun_out/r4_ahds_profrun.json 2> gpurun_out/r4_ahds_profrun.err
python bench.py > gpurun_out/r4_bench0.json 2> gpurun_out/r4_bench0.err

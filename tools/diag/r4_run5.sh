python -m pytest tests/test_gpu_conv.py tests/test_gpu_groupnorm.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4_run5_tests.log
python -m pytest tests/test_gpu_network_parity.py tests/test_gpu_sharded_step.py -x -q -s -m gpu 2>&1 | tail -40 >> gpurun_out/r4_run5_tests.log
python tools/diag/gn_bandwidth.py > gpurun_out/r4_gn_bandwidth2.txt 2>&1
python tools/bench_ahds.py --steps 10 --warmup 4 > gpurun_out/r4_ahds_a.json 2>/dev/null
python tools/bench_ahds.py --steps 10 --warmup 4 --proxy-group 4 > gpurun_out/r4_ahds_proxy4.json 2>/dev/null
python tools/bench_ahds.py --steps 10 --warmup 4 --proxy-group 2 > gpurun_out/r4_ahds_proxy2.json 2>/dev/null
GIP_MIN_CONV_TILES=32 python tools/bench_ahds.py --steps 10 --warmup 4 --proxy-group 4 > gpurun_out/r4_ahds_proxy4_old.json 2>/dev/null

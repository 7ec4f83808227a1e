#!/bin/bash
python -m pytest tests/test_gpu_raster_parity.py tests/test_gpu_pipeline.py tests/test_gpu_glue.py tests/test_gpu_ahds_step.py tests/test_gpu_headline_parity.py tests/test_gpu_sharded_step.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4_run44_tests.txt
python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | head -1 | cut -c1-300 >> gpurun_out/r4_run44_tests.txt

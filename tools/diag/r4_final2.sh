#!/bin/bash
# final verification of the round-4 tree: full GPU suite, default bench line, AHDS step profile
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r4_final2_tests.log
python bench.py > gpurun_out/r4_bench_final4.json 2> gpurun_out/r4_bench_final4.err
bash tools/prof_ahds.sh > gpurun_out/r4_prof_ahds4.log 2>&1

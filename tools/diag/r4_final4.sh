#!/bin/bash
# last verification of the round-4 tree: full GPU suite, smoke, default bench line
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r4_final4_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r4_final4_tests.log 2>&1
python bench.py > gpurun_out/r4_bench_final5.json 2> gpurun_out/r4_bench_final5.err

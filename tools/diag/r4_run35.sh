#!/bin/bash
python bench.py > gpurun_out/r4_bench_final3.json 2> gpurun_out/r4_bench_final3.err

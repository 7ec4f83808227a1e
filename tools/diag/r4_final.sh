python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r4_final_tests.log
python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err
bash tools/collect_profiles.sh > gpurun_out/r4_collect.log 2>&1

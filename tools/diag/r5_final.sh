#!/bin/bash
# round 5 closing run: full GPU suite, profile collection (kernel stats + PMC + config4 counters + AHDS step trace), the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests -q -m gpu > gpurun_out/r5/final_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5/final_tests.log
tail -6 gpurun_out/r5/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/final_smoke.log 2>&1; tail -2 gpurun_out/r5/final_smoke.log
bash tools/collect_profiles.sh > gpurun_out/r5_collect.log 2>&1
tail -12 gpurun_out/r5_collect.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r5/final_bench.json 2> gpurun_out/r5/final_bench.err
echo "bench rc $?"; cut -c1-300 gpurun_out/r5/final_bench.json

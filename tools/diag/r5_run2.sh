#!/bin/bash
# round 5, GPU run 2: full GPU suite on the new scan / gather / conv_in / quant-fold / wide-head attention; raster A/B vs the round-4 library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests -x -q -m gpu > gpurun_out/r5/run2_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5/run2_tests.log
tail -8 gpurun_out/r5/run2_tests.log
for rep in 1 2; do
for lib in libgip_raster_r4.so libgip_raster.so; do
  GIP_RASTER_LIB=$lib python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-ahds --no-trained --no-exact --no-config4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib', d['ms_per_step'], d['roofline']['stage_ms_instrumented'])" >> gpurun_out/r5/run2_ab.txt
done
done
cat gpurun_out/r5/run2_ab.txt
python tools/diag/raster_step_ops.py > gpurun_out/r5/run2_step_ops.txt 2>&1
python tools/bench_ahds.py --steps 10 --warmup 4 > gpurun_out/r5/run2_ahds.json 2> gpurun_out/r5/run2_ahds.err
cut -c1-600 gpurun_out/r5/run2_ahds.json

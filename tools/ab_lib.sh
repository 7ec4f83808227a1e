#!/bin/bash
# same-box A/B of two raster library builds: tools/ab_lib.sh <libA.so> <libB.so>   (files under gaussianip_amd/lib/)
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  for rep in 1 2; do
  GIP_RASTER_LIB=$lib python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-ahds --no-trained 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib', d['ms_per_step'], d['roofline']['stage_ms_instrumented'])"
  done
done
for lib in "$@"; do echo "$lib"; GIP_RASTER_LIB=$lib python tools/diag/trained_stages.py 2>/dev/null | grep num_rendered; done
GIP_RASTER_LIB=${@: -1} python -m pytest tests/test_gpu_raster_parity.py tests/test_gpu_headline_parity.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -2

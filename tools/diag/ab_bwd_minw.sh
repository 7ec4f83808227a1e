#!/bin/bash
# same-box A/B of render_backward's occupancy hint (runs on the GPU box): default build, then -DBWD_MINW=6 (80 VGPRs, 6 waves / SIMD)
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ahds --no-trained --no-exact --repeats 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['roofline']['stage_ms_instrumented']['render_bwd'], d['roofline']['stage_ms_instrumented']['render_fwd'])"; }
run default; run default
touch gaussianip_amd/csrc/render_backward.hip
make -C gaussianip_amd/csrc NOSLP_render_backward="-fno-slp-vectorize -DBWD_MINW=6" ../lib/libgip_raster.so > /dev/null 2>&1
run minw6; run minw6
touch gaussianip_amd/csrc/render_backward.hip
make -C gaussianip_amd/csrc NOSLP_render_backward="-fno-slp-vectorize -DBWD_MINW=6 -DBWD_GRID=24576" ../lib/libgip_raster.so > /dev/null 2>&1
run minw6_grid24k; run minw6_grid24k

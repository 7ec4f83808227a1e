#!/bin/bash
# Steady-state per-step kernel summary of the full AHDS training step (runs on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
mkdir -p $OUT
rm -rf /tmp/prof_ahds
rocprofv3 --kernel-trace -d /tmp/prof_ahds -o st -- python3 $GRAFT_REPO_ROOT/tools/bench_ahds.py --steps 6 --warmup 4 "$@" > $OUT/ahds_trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/analyze_db.py /tmp/prof_ahds/st_results.db gip_preprocess_kernel 70 > $OUT/ahds_step_summary.txt
python3 $GRAFT_REPO_ROOT/tools/dump_step.py /tmp/prof_ahds/st_results.db gip_preprocess_kernel > $OUT/ahds_step_trace.txt 2>&1
head -14 $OUT/ahds_step_summary.txt

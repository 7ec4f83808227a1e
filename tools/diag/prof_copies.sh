#!/bin/bash
# kernels + memory copies of one AHDS step (runs on the GPU box through gpurun): tools/diag/prof_copies.sh [from_us to_us]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cp
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_cp -o st -- python3 $GRAFT_REPO_ROOT/tools/bench_ahds.py --steps 6 --warmup 4 > /tmp/prof_cp.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/diag/copies_in_step.py /tmp/prof_cp/st_results.db gip_preprocess_kernel "$@" > $GRAFT_REPO_ROOT/gpurun_out/copies_in_step.txt 2>&1
head -100 $GRAFT_REPO_ROOT/gpurun_out/copies_in_step.txt

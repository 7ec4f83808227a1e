#!/bin/bash
# The pool's boxes differ by 5-8 % in sustained clock: take the step profile only on a box whose unprofiled step is below the
# threshold (ms), so that the committed summary and the quoted A/B numbers come from the same class of box.
# usage (through gpurun): tools/diag/prof_if_fast.sh 36.6
cd $GRAFT_REPO_ROOT
MS=$(python tools/bench_ahds.py --steps 10 --warmup 4 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
echo "unprofiled step: $MS ms (threshold $1)"
if python -c "import sys; sys.exit(0 if float('$MS') <= float('$1') else 1)"; then
  bash tools/prof_ahds.sh > /dev/null 2>&1
  echo "unprofiled step on this box: $MS ms" >> gpurun_out/profiles_new/ahds_step_summary.txt
  head -12 gpurun_out/profiles_new/ahds_step_summary.txt
else
  echo "slow box: no profile taken"
fi

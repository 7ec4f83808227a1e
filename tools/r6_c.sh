#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6c; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_sh_mfma.py::test_degree3_four_view_launch_set_against_the_oracle_at_100k_1024 tests/test_gpu_anpg_sensitivity.py "tests/test_gpu_sharded_step.py::test_config3_real_guidance_sharded_step_at_100k_1024" -q -m gpu -s 2>&1 | tail -60 > $OUT/tests.txt
cat $OUT/tests.txt | cut -c1-400

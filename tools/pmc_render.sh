#!/bin/bash
# PMC passes for the render kernels (separate passes; no trace domains combined with --pmc besides kernel-trace)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ahds --profile-iters 1 > $OUT/$TAG.log 2>&1 || true; }
TAG=p1; run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
TAG=p2; run SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
TAG=p3; run GRBM_GUI_ACTIVE FETCH_SIZE
TAG=p4; run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls -R $OUT | head -40

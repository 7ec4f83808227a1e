#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmc_nn
rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$TAG -- python3 $GRAFT_REPO_ROOT/tools/run_conv_once.py 12 640 640 64 > $OUT/$TAG.log 2>&1 || true; }
TAG=p1; run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_MFMA
TAG=p2; run SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM
TAG=p3; run GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL
TAG=p4; run SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACCUM_PREV
python3 - conv3x3 <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_nn/*/*/*_counter_collection.csv") + glob.glob("/tmp/pmc_nn/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print(k)
    for n, v in sorted(agg[k].items()):
        print("   %-32s %14.4g  (n=%d)" % (n, sum(v) / len(v), len(v)))
PY
grep -h conv3x3 /tmp/pmc_nn/p1/*/*kernel_trace.csv 2>/dev/null | head -3 | cut -c1-300

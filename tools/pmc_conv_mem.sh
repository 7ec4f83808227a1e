#!/bin/bash
# Texture-addresser / L1 / L2 counters of the implicit-GEMM convolution on its main shapes (what the LDS-DMA operand stream costs):
#   bash tools/pmc_conv_mem.sh            (on the GPU box; writes gpurun_out/profiles_new/conv_mem_pmc.txt)
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmc_mem
rm -rf $OUT; mkdir -p $OUT $GRAFT_REPO_ROOT/gpurun_out/profiles_new
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
grep -o -E "\b(TA|TCP|TD|TCC)_[A-Z0-9_]+" $OUT/avail.txt | sort -u > $GRAFT_REPO_ROOT/gpurun_out/profiles_new/avail_mem_counters.txt
SHAPES=("12 320 320 64" "4 128 128 512")
i=0
for shp in "${SHAPES[@]}"; do
  i=$((i+1))
  run() { timeout 75 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/s$i/$TAG -- python3 $GRAFT_REPO_ROOT/tools/run_conv_once.py $shp > $OUT/s$i.$TAG.log 2>&1 || true; }
  TAG=p1; run GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum
  TAG=p2; run TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
  TAG=p3; run TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
  TAG=p5; run SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
  TAG=p6; run TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
done
python3 - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/profiles_new/conv_mem_pmc.txt
import csv, glob, collections
names = {"s1": "conv3x3 320 -> 320 @ 64^2 x 12 (conv3x3_kernel<160>)", "s2": "conv3x3 128 -> 128 @ 512^2 x 4 (halo-resident tile)", "s3": "conv3x3 1280 -> 1280 @ 8^2 x 12 (split-K)"}
for s in ("s1", "s2"):
    agg = collections.defaultdict(list)
    dur = []
    for f in glob.glob("/tmp/pmc_mem/%s/*/*/*_counter_collection.csv" % s) + glob.glob("/tmp/pmc_mem/%s/*/*_counter_collection.csv" % s):
        for r in csv.DictReader(open(f)):
            if "conv3x3_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob("/tmp/pmc_mem/%s/p5/*/*_kernel_trace.csv" % s) + glob.glob("/tmp/pmc_mem/%s/p5/*_kernel_trace.csv" % s):
        for r in csv.DictReader(open(f)):
            if "conv3x3_kernel" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(names[s], " launch us (under the counters):", ["%.1f" % d for d in dur])
    for n, v in sorted(agg.items()):
        print("   %-40s %16.5g  (n=%d)" % (n, sum(v) / len(v), len(v)))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/profiles_new/conv_mem_pmc.txt
grep -l -i "error\|invalid\|not found" $OUT/*.log 2>/dev/null | head; grep -h -i "error\|invalid" $OUT/*.log | sort | uniq -c | head -10

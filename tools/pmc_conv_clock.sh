#!/bin/bash
# Clock of the halo-resident convolution under its timing ablations: TCP_GATE_EN1 (cycles the L1 is clocked, summed over the CUs)
# over the launch duration = the shader clock the kernel ran at.  Writes gpurun_out/profiles_new/conv_clock.txt
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmc_clk
rm -rf $OUT; mkdir -p $OUT $GRAFT_REPO_ROOT/gpurun_out/profiles_new
for ab in 0 2 3 4 7; do
  timeout 75 rocprofv3 --kernel-trace --pmc TCP_GATE_EN1_sum SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a$ab -- python3 $GRAFT_REPO_ROOT/tools/run_conv_once.py 4 128 128 512 $ab > $OUT/a$ab.log 2>&1 || true
done
python3 - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/profiles_new/conv_clock.txt
import csv, glob, collections
names = {0: "full kernel", 2: "no weight DMA", 3: "no DMA in the K loop", 4: "DMA only (no LDS reads / MFMA)", 7: "halo load + epilogue only"}
print("conv3x3 128 -> 128 @ 512^2 x 4, halo-resident kernel; 256 CUs, 1024 SIMDs")
for ab in (0, 2, 3, 4, 7):
    cnt = collections.defaultdict(list); dur = []
    for f in glob.glob("/tmp/pmc_clk/a%d/*/*_counter_collection.csv" % ab) + glob.glob("/tmp/pmc_clk/a%d/*_counter_collection.csv" % ab):
        for r in csv.DictReader(open(f)):
            if "conv3x3_kernel" in r["Kernel_Name"]:
                cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob("/tmp/pmc_clk/a%d/*/*_kernel_trace.csv" % ab) + glob.glob("/tmp/pmc_clk/a%d/*_kernel_trace.csv" % ab):
        for r in csv.DictReader(open(f)):
            if "conv3x3_kernel" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if not dur or not cnt:
        print("%-34s no data" % names[ab]); continue
    d = sum(dur) / len(dur)
    g = sum(cnt["TCP_GATE_EN1_sum"]) / max(len(cnt["TCP_GATE_EN1_sum"]), 1)
    m = sum(cnt["SQ_VALU_MFMA_BUSY_CYCLES"]) / max(len(cnt["SQ_VALU_MFMA_BUSY_CYCLES"]), 1)
    print("%-34s %7.1f us   clock %.2f GHz   matrix pipe busy %4.1f %% of the cycles" % (names[ab], d, g / 256 / d / 1e3, 100.0 * m / 1024 / max(g / 256, 1)))
PY

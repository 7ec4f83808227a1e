#!/bin/bash
# Calibrates what SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU read on streams whose issue rate is known from s_memtime stamps
# (tools/micro/valu_rate): runs on the GPU box through gpurun, writes gpurun_out/valu_rate_pmc.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf /tmp/vr_pmc
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/vr_pmc -- $GRAFT_REPO_ROOT/tools/micro/valu_rate > $OUT/valu_rate_under_pmc.txt 2>&1
python3 - <<PY > $OUT/valu_rate_pmc.txt
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/vr_pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:40], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# kernel, grid threads, workgroup -> counters (mean over the 5 launches), derived ratios")
for key in sorted(rows):
    c = {n: sum(v) / len(v) for n, v in rows[key].items()}
    insts, act = c.get("SQ_INSTS_VALU", 0), c.get("SQ_ACTIVE_INST_VALU", 0)
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    print("%-42s grid %8d wg %5d  INSTS_VALU %.4g  ACTIVE_INST_VALU %.4g  ACTIVE/INSTS %.3f  launch cycles %.4g  4*INSTS/(1024*cyc) %.3f  ACTIVE/(1024*cyc) %.3f  ACTIVE*4/(1024*cyc) %.3f  SQ_BUSY_CYCLES %.4g SQ_WAVE_CYCLES %.4g" % (
        key[0], key[1], key[2], insts, act, act / max(insts, 1), cyc, 4 * insts / max(1024 * cyc, 1), act / max(1024 * cyc, 1), 4 * act / max(1024 * cyc, 1), c.get("SQ_BUSY_CYCLES", 0), c.get("SQ_WAVE_CYCLES", 0)))
PY
cat $OUT/valu_rate_pmc.txt

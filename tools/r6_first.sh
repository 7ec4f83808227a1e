#!/bin/bash
# round 6, first GPU call: (1) cost of a dependent stage boundary: graph node vs grid barrier; (2) segment length 64 / 128 / 256 on the
# headline step and on configs[4]; (3) HBM counters of the 1M-Gaussian render kernels by views per launch set.
# The segment-length variants are library builds, made beforehand (they are not kept in the tree):
#   for S in 128 256; do make -C gaussianip_amd/csrc OUT=/tmp/seg$S HIPCC="/opt/rocm/bin/hipcc -DGIP_SEGMENT=$S" /tmp/seg$S/libgip_raster.so
#   cp /tmp/seg$S/libgip_raster.so gaussianip_amd/lib/libgip_raster_seg$S.so; done
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $OUT
timeout 120 tools/micro/grid_barrier 64 > $OUT/grid_barrier.txt 2>&1
for lib in libgip_raster.so libgip_raster_seg128.so libgip_raster_seg256.so libgip_raster.so; do
  GIP_RASTER_LIB=$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ahds --no-proxy 2>$OUT/bench_$lib.err | tail -1 > $OUT/bench_$lib.json
  python - <<PY
import json
d = json.loads(open("$OUT/bench_$lib.json").readline())
c4 = d.get("config4", {})
print("$lib", "step", d["ms_per_step"], "exact", d.get("exact_lists", {}).get("ms_per_step"), "trained", d.get("trained_state", {}).get("ms_per_step"),
      "c4 fwd", c4.get("forward_ms_per_set"), "fwd+bwd", c4.get("forward_backward_ms_per_set"),
      {k: v["ms"] for k, v in c4.get("stages_instrumented", {}).items()}, d["roofline"].get("stage_ms_instrumented"))
PY
done 2>&1 | tee $OUT/seg_ab.txt
cd /tmp && export TMPDIR=/tmp
for V in 1 4 12; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/c4_${C}_v$V -- python3 $GRAFT_REPO_ROOT/tools/run_config4_once.py 2 $V > $OUT/c4_${C}_v$V.log 2>&1
  done
done
export GIP_RASTER_LIB=libgip_raster_seg256.so
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/c4_${C}_v12_seg256 -- python3 $GRAFT_REPO_ROOT/tools/run_config4_once.py 2 12 > $OUT/c4_${C}_v12_seg256.log 2>&1
done
python3 - <<PY
import csv, glob, collections
out = "$OUT"
for tag in ("v1", "v4", "v12", "v12_seg256"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(out + "/c4_%s_%s/*/*_counter_collection.csv" % (C, tag)):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "gip_" in k:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== views per launch set:", tag)
    for k in sorted(agg):
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        print("%-34s fetch(x2) %14d write %14d" % (k, c.get("FETCH_SIZE", 0) * 2048, c.get("WRITE_SIZE", 0) * 1024))
PY

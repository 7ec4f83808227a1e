#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats of the default bench + PMC passes for HBM traffic.
# Summaries land in gpurun_out/profiles_new/ ; copy what should be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ahds --no-trained --no-exact --no-config4 --repeats 1 > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ahds --no-trained --no-exact --no-config4 --repeats 1 --profile-iters 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ahds --no-trained --no-exact --no-config4 --repeats 1 --profile-iters 1 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ahds --no-trained --no-exact --no-config4 --repeats 1 --profile-iters 1 > $OUT/pmc_sq.log 2>&1
# vector-instruction class mix of the render kernels (what roofline_valu prices: transcendental 8.1 cycles, everything else 2.24)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/pmc_mix -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ahds --no-trained --no-exact --no-config4 --repeats 1 --profile-iters 1 > $OUT/pmc_mix.log 2>&1
python3 - <<PY
import csv, glob, json, collections, os
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mix"):
    for f in glob.glob(out + "/" + d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "gip_" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines, traffic, pmc = [], {}, {}
for k in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    # gfx950: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X guide, HBM section)
    fetch = c.get("FETCH_SIZE", 0.0) * 1024 * 2
    write = c.get("WRITE_SIZE", 0.0) * 1024
    stage = k.replace("gip_", "").replace("_kernel", "").replace("render_forward", "render_fwd").replace("render_backward", "render_bwd").replace("gather_backward<1>", "gather_bwd").replace("gather_backward", "gather_bwd")
    traffic[stage] = int(fetch + write)
    pmc[stage] = dict(c, hbm_fetch_bytes=int(fetch), hbm_write_bytes=int(write))
    lines.append("%-34s HBM bytes/launch ~ %12d (fetch x2 %12d + write %12d)  " % (k, fetch + write, fetch, write) + "  ".join("%s=%.4g" % kv for kv in sorted(c.items())))
import sys
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
import bench
sha = bench.raster_source_hash()
pmc["_build"] = {"raster_source_sha16": sha, "git": os.environ.get("GIP_GIT", "unknown"),
                 "note": "sha256[:16] of the rasterizer's kernel sources + headers + Makefile (bench.raster_source_hash); bench.py emits roofline.traffic / roofline_valu only when it equals the sources it was built from"}
lines.insert(0, "# counters of raster sources sha256[:16] = %s, git %s" % (sha, os.environ.get("GIP_GIT", "unknown")))
open(out + "/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(out + "/traffic.json", "w"), indent=1)
json.dump(pmc, open(out + "/pmc.json", "w"), indent=1)
print("\n".join(lines))
PY
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
# BASELINE configs[4] (1M Gaussians, one 12-view launch set): kernel stats + HBM counters where the roofline is meaningful
if [ -z "$SKIP_CONFIG4" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats -- python3 $GRAFT_REPO_ROOT/tools/run_config4_once.py 4 > $OUT/c4_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c4_fetch -- python3 $GRAFT_REPO_ROOT/tools/run_config4_once.py 2 > $OUT/c4_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c4_write -- python3 $GRAFT_REPO_ROOT/tools/run_config4_once.py 2 > $OUT/c4_write.log 2>&1
  cp $OUT/c4_stats/*/*_kernel_stats.csv $OUT/config4_kernel_stats.csv
  python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("c4_fetch", "c4_write"):
    for f in glob.glob(out + "/" + d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "gip_" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for r in csv.DictReader(open(out + "/config4_kernel_stats.csv")):
    k = r["Name"].split("(")[0].replace("void ", "")
    if "gip_" in k:
        dur[k] = float(r["AverageNs"]) / 1e3
lines = ["# BASELINE configs[4]: 1M Gaussians, 1024^2, one 12-view launch set; gfx950 corrections: FETCH_SIZE x 2, KiB units",
         "# %-32s %10s %14s %14s %10s" % ("kernel", "avg us", "fetch bytes", "write bytes", "HBM GB/s")]
for k in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    fetch, write = c.get("FETCH_SIZE", 0.0) * 1024 * 2, c.get("WRITE_SIZE", 0.0) * 1024
    us = dur.get(k, 0.0)
    lines.append("%-34s %10.1f %14d %14d %10.1f" % (k, us, fetch, write, (fetch + write) / (us * 1e-6) / 1e9 if us else 0.0))
open(out + "/pmc_config4_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
fi
# full AHDS training step (raster + VAE + ControlNet + U-Net + Adam): steady-state per-step kernel summary
if [ -n "$SKIP_AHDS" ]; then tail -1 $OUT/bench_stats.log | cut -c1-600; exit 0; fi
rocprofv3 --kernel-trace -d /tmp/prof_ahds -o st -- python3 $GRAFT_REPO_ROOT/tools/bench_ahds.py --steps 6 --warmup 4 > $OUT/ahds_trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/analyze_db.py /tmp/prof_ahds/st_results.db gip_preprocess_kernel 60 > $OUT/ahds_step_summary.txt
python3 $GRAFT_REPO_ROOT/tools/dump_step.py /tmp/prof_ahds/st_results.db gip_preprocess_kernel > $OUT/ahds_step_trace.txt 2>&1
head -16 $OUT/ahds_step_summary.txt
tail -1 $OUT/bench_stats.log | cut -c1-300

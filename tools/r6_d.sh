#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6d; mkdir -p $OUT
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt
tail -3 $OUT/bench_time.txt; tail -5 $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").readline())
print("raster", d["ms_per_step"], "ahds", d.get("ahds_ms_per_step"), d["ahds"].get("trained_state"), d["ahds"].get("error"))
print(json.dumps(d["config4"].get("refine")), json.dumps(d["config4"].get("stage3")), d["config4"].get("refine_error"))
print(json.dumps(d["trained_state"]))
print(json.dumps(d["ahds"].get("config3_proxy")))
PY
timeout 1500 python -m pytest tests/test_gpu_refine.py -x -q -m gpu -s 2>&1 | tail -40 | cut -c1-300

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests -q -m gpu > gpurun_out/r5/final2_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5/final2_tests.log
tail -4 gpurun_out/r5/final2_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
python bench.py > gpurun_out/r5/final2_bench.json 2> gpurun_out/r5/final2_bench.err
echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/final2_bench.json').readline())
print(d['ms_per_step'], d['roofline']['frac'], d['ahds']['ms_per_step'], d['ahds']['value'], d['ahds'].get('config3_proxy',{}).get('group_of_4'))
PY

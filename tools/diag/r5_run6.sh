#!/bin/bash
# round 5, GPU run 6: GroupNorm backward apply software-pipelined (A/B against the same build without it), scan with the parallel block role
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_groupnorm.py tests/test_gpu_raster_parity.py tests/test_gpu_config4.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r5/run6_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5/run6_tests.log
tail -4 gpurun_out/r5/run6_tests.log
for rep in 1 2 3; do
for lib in libgip_nn_nopipe.so libgip_nn.so; do
  GIP_NN_LIB=$lib python tools/exp_vae_time.py 2>/dev/null | tail -1 | sed "s/^/$lib /" >> gpurun_out/r5/run6_ab_vae.txt
done
done
cat gpurun_out/r5/run6_ab_vae.txt
python tools/diag/gn_bandwidth.py > gpurun_out/r5/run6_gn_bandwidth.txt 2>&1; tail -15 gpurun_out/r5/run6_gn_bandwidth.txt
python - <<'PY' > gpurun_out/r5/run6_config4.json 2> gpurun_out/r5/run6_config4.err
import json, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
print(json.dumps(bench.measure_config4(torch.device("cuda"))))
PY
python -c "
import json
d=json.load(open('gpurun_out/r5/run6_config4.json'))
print({k:v['ms'] for k,v in d['stages_instrumented'].items()}, d['forward_ms_per_set'], d['forward_backward_ms_per_set'], d['whole_step_frac_of_8TBs'])"

python tools/bench_orbit.py > gpurun_out/r4_orbit.txt 2>&1
python tools/bench_refine.py > gpurun_out/r4_refine.txt 2>&1
python tools/bench_stage3.py > gpurun_out/r4_stage3.txt 2>&1

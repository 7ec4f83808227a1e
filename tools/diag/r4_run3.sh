python tools/diag/guidance_determinism.py > gpurun_out/r4_determinism.txt 2>&1
python tools/diag/gn_bandwidth.py > gpurun_out/r4_gn_bandwidth.txt 2>&1

python tools/diag/denoise_bisect.py 2 > gpurun_out/r4_bisect_2streams.txt 2>&1
GIP_GUIDANCE_STREAMS=1 python tools/diag/denoise_bisect.py 2 > gpurun_out/r4_bisect_1stream.txt 2>&1
python tools/diag/denoise_bisect.py 1 > gpurun_out/r4_bisect_b1.txt 2>&1
python tools/diag/denoise_bisect.py 4 > gpurun_out/r4_bisect_b4.txt 2>&1

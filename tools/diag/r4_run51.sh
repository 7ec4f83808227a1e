#!/bin/bash
# is the ControlNet || U-Net encoder overlap alive inside the captured denoise graph?  (the profiled step shows the two queues one after the other)
bash tools/ab_ahds.sh "GIP_X=1" "GIP_GUIDANCE_STREAMS=1" "GIP_GRAPH_DENOISE=0" "GIP_GRAPH_DENOISE=0 GIP_GUIDANCE_STREAMS=1" > gpurun_out/r4_ab_streams.txt 2>&1

#!/bin/bash
# same-box A/B: render_bwd's 64-lane sums on the matrix pipe (BWD_MFMA=1) against the permlane / DPP folds
bash tools/ab_lib.sh libgip_raster_base.so libgip_raster_mfma.so libgip_raster_base.so libgip_raster_mfma.so > gpurun_out/r4_ab_mfma.txt 2>&1

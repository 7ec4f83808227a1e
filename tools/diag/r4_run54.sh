#!/bin/bash
timeout 600 python -u tools/diag/graph_with_rccl.py > gpurun_out/r4_graph_rccl.txt 2> gpurun_out/r4_graph_rccl.err
echo "rc=$?" >> gpurun_out/r4_graph_rccl.txt
tail -5 gpurun_out/r4_graph_rccl.err | grep -v amdgpu >> gpurun_out/r4_graph_rccl.txt

#!/bin/bash
python -m pytest tests/test_gpu_glue.py -x -q -m gpu -k activations 2>&1 | grep -v Warning | tail -40 > gpurun_out/r4_glue_tests3.txt
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_glue.py::test_fused_activations_equal_the_getters_and_their_autograd 2>&1 | tail -8 >> gpurun_out/r4_glue_tests3.txt

#!/bin/bash
# last verification of the round-4 tree: full GPU suite + smoke
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r4_final3_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r4_final3_tests.log 2>&1

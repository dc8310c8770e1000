#!/bin/bash
# round-4 closing session: the whole GPU suite on the final tree, then the driver's bench command with its kernel traces
o=gpurun_out/r4_finalbench; mkdir -p $o
python -m pytest tests -m gpu -q > $o/gpu_tests.log 2>&1; tail -2 $o/gpu_tests.log
bash tools/r4_final_bench.sh 2>&1 | tail -12

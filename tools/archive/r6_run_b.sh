#!/bin/bash
# Round 6, GPU session B: the Winograd GPU test that failed in session A with its traceback; Adam with device step counts.
o=gpurun_out/r6b; mkdir -p $o
timeout 600 python -m pytest tests/test_winograd.py -m gpu -q 2>&1 | tail -n 60 | tee $o/wino.txt
timeout 600 python -m pytest tests/test_adam.py -m gpu -q 2>&1 | tail -n 15 | tee $o/adam.txt
echo SESSION_B_DONE

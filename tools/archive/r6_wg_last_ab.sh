#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
timeout 600 python tools/ab_wprep.py wp_head product 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6p/wprep_32bit.txt
timeout 900 python -m pytest tests/test_weight_prep.py tests/test_winograd.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -2

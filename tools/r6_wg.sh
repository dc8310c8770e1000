#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
timeout 600 python tools/ab_k1_epilogue.py k1_h4 product k1_h4 product 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6p/k1_x2_phases.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize_oracle.py tests/test_resblock_fused.py tests/test_gpu_fullsize_properties.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r6p/k1_tests5.txt

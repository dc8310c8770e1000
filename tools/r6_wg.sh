#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
timeout 600 python tools/wf_variants.py wf_h product 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6p/wf_quad_factors.txt
timeout 900 python -m pytest tests/test_winograd.py tests/test_gpu_fullsize_oracle.py tests/test_styled_fused.py -m gpu -x -q -k "wino or Wino or styled" 2>&1 | tail -3 | tee gpurun_out/r6p/wino_tests7.txt

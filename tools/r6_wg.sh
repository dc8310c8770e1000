#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
timeout 600 python tools/wf_variants.py wf_zp product 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6p/wf_xs_uniform.txt
timeout 900 python -m pytest tests/test_winograd.py tests/test_gpu_fullsize_oracle.py tests/test_styled_fused.py tests/test_modconv.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r6p/wino_tests5.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
timeout 600 python tools/ab_k2.py k2_head product k2_head product 2>&1 | grep -v amdgpu.ids | grep "shape\|bias_act" | tee gpurun_out/r6p/k2_incremental.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
(timeout 600 python tools/ab_k2.py k1_head product k1_head product 2>&1 | grep -v amdgpu.ids | grep "shape\|blur"
 timeout 600 python tools/ab_k1_epilogue.py k1_head product k1_head product 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/r6p/k1_buffer_loads.txt

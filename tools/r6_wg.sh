#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
timeout 600 python tools/ab_k1_epilogue.py k1_h3 product k1_h3 product 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6p/k1_uniform_updown.txt

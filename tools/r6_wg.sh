#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6p
timeout 600 python tools/kb_k1.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6p/kb_k1.txt

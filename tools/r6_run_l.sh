#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6l
timeout 900 python tools/ab_conv.py late_base late_1x1 late_base late_1x1 "--1x1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6l/late_1x1.txt
echo SESSION_L_DONE

#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6l
timeout 900 python tools/ab_conv.py wg_spread2 ig_spread1 ig_spread2 ig_spread3 --op=fwd "--s2 " "--Dp s2" "--Dp3 s2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6l/ig_spread.txt

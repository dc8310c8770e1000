#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6l
timeout 1500 python tools/ab_conv.py late_base late_a late_b late_c --op=fwd --op=wgrad 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6l/late_all.txt
echo SESSION_L_DONE

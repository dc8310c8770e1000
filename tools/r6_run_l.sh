#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6l
timeout 600 python tools/wf_variant_check.py wf_sched2 2>&1 | tail -1
timeout 900 python tools/wf_variants.py product wf_sched2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6l/wf_sched2.txt
echo SESSION_L_DONE

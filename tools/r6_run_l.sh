#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6l
timeout 900 python tools/wf_variants.py product wg_h wg_i wg_j wg_k wg_l 2>&1 | grep -v amdgpu.ids | awk -F'|' '{n=split($0,a,"|"); out=substr(a[1],1,13); for(i=1;i<=n;i++){ if (match(a[i], /wgrad [0-9.]+ ms \([0-9.]+\)/)) out=out " | " substr(a[i],RSTART,RLENGTH)}; print out}' | tee gpurun_out/r6l/wg_orders2.txt
echo SESSION_L_DONE

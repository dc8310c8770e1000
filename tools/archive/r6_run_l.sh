#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6l
timeout 600 python tools/wf_variant_check.py wf_mbl2 2>&1 | tail -1
for v in wf_mbl0 wf_mbl2; do timeout 300 python tools/wf_fixed_cost.py $v 2>&1 | grep -v amdgpu.ids; done
timeout 900 python tools/wf_variants.py product wf_mbl0 wf_mbl2 wf_mbl8 2>&1 | grep -v amdgpu.ids | awk -F'|' '{n=split($0,a,"|"); out=substr(a[1],1,13); for(i=1;i<=n;i++){ if (match(a[i], /conv [0-9.]+ ms \([0-9.]+\)/)) out=out " | " substr(a[i],RSTART,RLENGTH)}; print out}' | tee gpurun_out/r6l/wf_mbloop.txt
echo SESSION_L_DONE

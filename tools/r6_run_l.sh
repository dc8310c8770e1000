#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6l
cp swapping_autoencoder_pytorch_amd/csrc/libsae_hip.so tools/variants/prod_copy.so
timeout 600 python tools/wf_variant_check.py prod_copy 2>&1 | tail -1
timeout 900 python tools/wf_variants.py product wc_a wc_b wc_c wc_d wc_e 2>&1 | grep -v amdgpu.ids | awk -F'|' '{n=split($0,a,"|"); out=substr(a[1],1,13); for(i=1;i<=n;i++){ if (match(a[i], /conv [0-9.]+ ms \([0-9.]+\)/)) out=out " | " substr(a[i],RSTART,RLENGTH)}; print out}' | tee gpurun_out/r6l/wc_orders.txt
echo SESSION_L_DONE

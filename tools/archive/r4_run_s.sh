#!/bin/bash
# round-4 session S: conv_igemm_ws_kernel after the role split (95 registers: two workgroups per CU whatever the placement)
F="--128@256 --256@128 --64@64"
SAE_WS=0 timeout 15 python tools/ab_conv.py tuning $F --op=fwd --op=dgrad 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_ws0b.txt
SAE_WS=1 timeout 15 python tools/ab_conv.py tuning $F --op=fwd --op=dgrad 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_ws1b.txt
paste -d'|' <(cut -c1-38 gpurun_out/r4_ws0b.txt) <(cut -c28-38 gpurun_out/r4_ws1b.txt)
echo DONE

#!/bin/bash
# round-4 session R: conv_igemm_ws_kernel (producer / consumer waves) -- bit identity on the GPU, then same-box timing
timeout 45 python -m pytest tests/test_ws_gather.py -m gpu -x -q 2>&1 | tail -3
F="--128@256 --256@128 --64@64"
SAE_WS=0 timeout 15 python tools/ab_conv.py tuning $F --op=fwd --op=dgrad 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_ws0.txt
SAE_WS=1 timeout 15 python tools/ab_conv.py tuning $F --op=fwd --op=dgrad 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_ws1.txt
paste -d'|' <(cut -c1-38 gpurun_out/r4_ws0.txt) <(cut -c28-38 gpurun_out/r4_ws1.txt)
echo DONE

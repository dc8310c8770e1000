#!/bin/bash
# Round 5, GPU session B: per-layer A/B of the Winograd route (eligibility rule), and parity with the route ON.
o=gpurun_out/r5b; mkdir -p $o
rm -f gpurun_out/network_parity*.jsonl
timeout 600 python -m pytest tests/test_gpu_network_parity.py -q 2>&1 | tail -5 > $o/netparity_default.log; tail -3 $o/netparity_default.log
mv gpurun_out/network_parity_tensors.jsonl $o/network_parity_tensors_default.jsonl; mv gpurun_out/network_parity.jsonl $o/network_parity_default.jsonl
SAE_WINOGRAD=1 timeout 900 python -m pytest tests/test_gpu_network_parity.py tests/test_gpu_parity.py tests/test_resblock_fused.py tests/test_styled_fused.py -m gpu -q 2>&1 | tail -30 > $o/parity_winograd_on.log; tail -8 $o/parity_winograd_on.log
mv gpurun_out/network_parity_tensors.jsonl $o/network_parity_tensors_winograd.jsonl; mv gpurun_out/network_parity.jsonl $o/network_parity_winograd.jsonl
timeout 900 python tools/wino_ab.py --preset church256 > $o/wino_ab_church256.json 2> $o/wino_ab.err; tail -3 $o/wino_ab.err; tail -4 $o/wino_ab_church256.json
timeout 600 python tools/wino_ab.py --preset ffhq512 > $o/wino_ab_ffhq512.json 2>> $o/wino_ab.err; tail -4 $o/wino_ab_ffhq512.json
echo SESSION_B_DONE

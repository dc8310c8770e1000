#!/bin/bash
# Round 6, GPU session D: parity hardening (full-size Dpatch network, whole D call + G call vs the double restatement, fused blocks
# vs the oracle, post-activation Winograd cases, the free-run flip assertion) and the multi-rank instrumentation rehearsed with two
# ranks on the one GPU.
o=gpurun_out/r6d; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_step_parity.py tests/test_resblock_fused.py tests/test_styled_fused.py -m gpu -q 2>&1 | tail -n 25 | tee $o/parity_new.txt
timeout 1500 python -m pytest tests/test_gpu_network_parity.py -m gpu -q 2>&1 | tail -n 25 | tee $o/network_parity.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize_oracle.py -m gpu -q -k "winograd" 2>&1 | tail -n 25 | tee $o/wino_fullsize.txt
timeout 900 python -m pytest tests/test_ddp_fullmodel.py tests/test_gpu_allreduce.py -m gpu -q 2>&1 | tail -n 15 | tee $o/ddp.txt
cp gpurun_out/step_parity_fullsize.jsonl gpurun_out/network_parity.jsonl gpurun_out/ddp_fullmodel_one_gpu.txt $o/ 2>/dev/null
timeout 900 python bench.py --gpus 2 --same-device --steps 4 --warmup 2 --alt-steps 0 --kernel-steps 0 --no-kernel-timing --alt-streams-steps 2 > $o/bench_two_ranks_one_gpu.json 2> $o/bench_two_ranks_one_gpu.err
tail -c 1500 $o/bench_two_ranks_one_gpu.json; tail -n 5 $o/bench_two_ranks_one_gpu.err
echo SESSION_D_DONE

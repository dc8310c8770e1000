#!/bin/bash
# Round 6, GPU session E: the fused ResBlock against the oracle (traceback), gemm split across workgroups, the two-rank rehearsal
# of bench.py with stacks if it sticks.
o=gpurun_out/r6e; mkdir -p $o
timeout 600 python -m pytest tests/test_resblock_fused.py tests/test_gpu_step_parity.py -m gpu -q 2>&1 | grep -v "^E   .*array\|amdgpu.ids" | tail -n 60 | cut -c1-400 | tee $o/resblock.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k gemm 2>&1 | tail -n 5 | tee $o/gemm.txt
export SAE_BENCH_STACKS_AFTER_S=150
for preset in tiny32 church256; do
timeout 300 python bench.py --preset $preset --gpus 2 --same-device --steps 4 --warmup 2 --alt-steps 0 --kernel-steps 0 --no-kernel-timing --alt-streams-steps 2 --no-cpu-baseline > $o/two_ranks_$preset.json 2> $o/two_ranks_$preset.err
echo "rc=$?"; tail -c 1200 $o/two_ranks_$preset.json; grep -v "amdgpu.ids\|hostname" $o/two_ranks_$preset.err | head -n 120 | cut -c1-300
done
echo SESSION_E_DONE

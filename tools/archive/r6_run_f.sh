#!/bin/bash
# Round 6, GPU session F: streaming blur (bit-identity + A/B), fused ResBlock vs oracle, gloo device all-reduce probe, the two-rank
# rehearsal of bench.py with the all-reduce staged through the host.
o=gpurun_out/r6f; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_resblock_fused.py -m gpu -q -k "streaming or upfirdn or against_the_oracle or k1 or epilogue" 2>&1 | grep -v "amdgpu.ids" | tail -n 12 | cut -c1-300 | tee $o/k1_tests.txt
python tools/ab_k1_stream.py 2>/dev/null | tee $o/ab_k1_stream.txt
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 tools/probe/gloo_device_allreduce_probe.py 2>&1 | grep -v "amdgpu.ids\|hostname\|OMP_NUM\|\*\*\*\*" | tail -n 12 | tee $o/gloo_probe.txt
export SAE_BENCH_STACKS_AFTER_S=200
for preset in tiny32 church256; do
timeout 400 python bench.py --preset $preset --gpus 2 --same-device --steps 4 --warmup 2 --alt-steps 0 --kernel-steps 0 --no-kernel-timing --alt-streams-steps 2 --no-cpu-baseline > $o/two_ranks_$preset.json 2> $o/two_ranks_$preset.err
echo "rc=$?"; tail -c 1600 $o/two_ranks_$preset.json; grep -v "amdgpu.ids\|hostname" $o/two_ranks_$preset.err | grep -i "error\|Timeout\|File" | head -n 30 | cut -c1-300
done
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
python bench.py $B 2>/dev/null | tail -n 1 | cut -c1-200 | tee $o/bench_quick.txt
echo SESSION_F_DONE

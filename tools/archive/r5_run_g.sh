#!/bin/bash
# Round 5, GPU session G: same-box timing of csrc/winograd_fused.hip variants (schedule: last point after the barrier + transform
# spread over four groups = product; the previous commit; weights by LDS-DMA), then the step.
o=gpurun_out/r5g; mkdir -p $o
timeout 600 python tools/wf_variants.py product wf_prev wf_dma 2>&1 | grep -v amdgpu.ids | tee $o/wf_variants.txt
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
python bench.py $B 2>/dev/null | line product | tee -a $o/wf_variants.txt
python tools/bench_variant.py wf_prev $B 2>/dev/null | line wf_prev | tee -a $o/wf_variants.txt
python tools/bench_variant.py wf_dma $B 2>/dev/null | line wf_dma | tee -a $o/wf_variants.txt
python bench.py $B 2>/dev/null | line product | tee -a $o/wf_variants.txt
echo SESSION_G_DONE

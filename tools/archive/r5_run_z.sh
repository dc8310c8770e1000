#!/bin/bash
# Round 5, GPU session Z: select-free staging of interior tiles / chunks in the stride-2 gather and weight gradient -- per-shape A/B
# against HEAD's kernels (bit-identity checked), quad-path parity on the GPU, step A/B.
o=gpurun_out/r5z; mkdir -p $o
python tools/ab_conv.py base fast --s2 --op=fwd 2>&1 | grep -v amdgpu | tee $o/ab_s2_fwd.txt
python tools/ab_conv.py base fast --s2 --op=wgrad 2>&1 | grep -v amdgpu | tee $o/ab_s2_wgrad.txt
timeout 600 python -m pytest tests/test_quad_paths.py tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_modconv.py -m gpu -q 2>&1 | tail -n 3 | tee $o/gpu_subset.txt
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
for i in 1 2; do
python tools/bench_variant.py base $B 2>/dev/null | line base | tee -a $o/step_ab.txt
python bench.py $B 2>/dev/null | line product | tee -a $o/step_ab.txt
done
echo SESSION_Z_DONE

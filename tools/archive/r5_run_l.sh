#!/bin/bash
# Round 5, GPU session L: operand reads pinned ahead of their MFMAs in the transposed gather (tr2) and the 3x3 weight gradient
# (compile-time chunk width), tr2 weight image split for ds_read2st64 -- per-shape A/B (bit-identity checked), step A/B, kernel parity.
o=gpurun_out/r5l; mkdir -p $o
python tools/ab_conv.py nopin pin pin_noas pin_l1 pin_l3 --s2 --op=dgrad 2>&1 | grep -v amdgpu | tee $o/ab_tr2.txt
python tools/ab_conv.py nopin pin --op=wgrad 2>&1 | grep -v amdgpu | tee $o/ab_wgrad.txt
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
python tools/bench_variant.py nopin $B 2>/dev/null | line nopin | tee $o/step_ab.txt
python bench.py $B 2>/dev/null | line product | tee -a $o/step_ab.txt
python tools/bench_variant.py nopin $B 2>/dev/null | line nopin | tee -a $o/step_ab.txt
python bench.py $B 2>/dev/null | line product | tee -a $o/step_ab.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_quad_paths.py tests/test_modconv.py -m gpu -q 2>&1 | tail -n 4 | tee $o/gpu_subset.txt
echo SESSION_L_DONE

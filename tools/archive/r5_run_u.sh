#!/bin/bash
# Round 5, GPU session U: the one-kernel Winograd forward walking several tile blocks per workgroup (last chunk of a block stages the
# first chunk of the next): per-shape timing against the one-block form and HEAD's kernel, parity, step A/B.
o=gpurun_out/r5u; mkdir -p $o
python tools/wf_variants.py wf_head wf_bpw1 wf_bpw2 product wf_bpw8 2>&1 | grep -v amdgpu | tee $o/wf_variants.txt
timeout 600 python -m pytest tests/test_winograd.py -m gpu -q 2>&1 | tail -n 3 | tee $o/wino_gpu_tests.txt
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
for i in 1 2; do
python tools/bench_variant.py wf_head $B 2>/dev/null | line wf_head | tee -a $o/step_ab.txt
python bench.py $B 2>/dev/null | line product | tee -a $o/step_ab.txt
done
echo SESSION_U_DONE

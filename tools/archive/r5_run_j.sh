#!/bin/bash
# Round 5, GPU session J: the packed-transform variant of the one-kernel Winograd forward (SAE_WF_PK): instruction semantics, parity, timing.
o=gpurun_out/r5j; mkdir -p $o
./tools/probe/pkasm_probe | tee $o/pkasm_probe.txt
python tools/wf_variant_check.py wf_pk 2>&1 | grep -v amdgpu | tail -3 | tee $o/wf_pk_check.txt
python tools/wf_variants.py product wf_pk 2>&1 | grep -v amdgpu | tee $o/wf_variants.txt
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
python bench.py $B 2>/dev/null | line product | tee -a $o/wf_variants.txt
python tools/bench_variant.py wf_pk $B 2>/dev/null | line wf_pk | tee -a $o/wf_variants.txt
echo SESSION_J_DONE

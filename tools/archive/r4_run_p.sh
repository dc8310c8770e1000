#!/bin/bash
# round-4 session P: whole-step A/B on one box -- vA = current defaults, vB = 1x1 gather at the compiler's own occupancy,
# vA with the old plans (SAE_WGRAD_SLICE_MODEL=0 SAE_IGEMM_SPLIT_MODEL=0); two rounds
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing"
for r in 1 2; do
  for v in vA vB; do
    python tools/bench_variant.py $v $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"
  done
  SAE_WGRAD_SLICE_MODEL=0 SAE_IGEMM_SPLIT_MODEL=0 python tools/bench_variant.py vA $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vA-oldplans', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"
done
echo DONE

#!/bin/bash
# round-4 session Q: s_setprio during the MFMA phase, per kernel (variants q0 / g2 / gw1 / gw2 / gw3), isolated and in the step
python tools/ab_conv.py q0 g2 gw1 gw2 gw3 --128@256 --256@128 --512@64 "--512@32 n40" --@257 2>&1 | grep -v amdgpu.ids
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing"
for r in 1 2; do
  for v in q0 gw2 gw3; do
    python tools/bench_variant.py $v $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"
  done
done
echo DONE

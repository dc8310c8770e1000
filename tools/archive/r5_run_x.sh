#!/bin/bash
# Round 5, GPU session X: does the host block in the rehearsal's stream wait / launch?  And where does the host time of a step go
# with and without the rehearsal (py-spy is not here: wall-clock of train_one_step's host side through the bench's own medians).
o=gpurun_out/r5x; mkdir -p $o
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
export SAE_RING_REHEARSAL_KERNEL=persistent:32:20 SAE_RING_REHEARSAL_HOST_TIMES=1
SAE_ALLREDUCE_BUCKET_MB=512 python bench.py $B --force-allreduce --ring-rehearsal 8 2>$o/host_own.txt | line "own" | tee -a $o/host_times.txt
grep "host side" $o/host_own.txt | tail -n 2 | tee -a $o/host_times.txt
SAE_ALLREDUCE_BUCKET_MB=512 SAE_RING_REHEARSAL_STREAM=launch python bench.py $B --force-allreduce --ring-rehearsal 8 2>$o/host_launch.txt | line "launch" | tee -a $o/host_times.txt
grep "host side" $o/host_launch.txt | tail -n 2 | tee -a $o/host_times.txt
echo SESSION_X_DONE

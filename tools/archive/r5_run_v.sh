#!/bin/bash
# Round 5, GPU session V: the ring rehearsal with ONE persistent kernel per bucket (the shape of RCCL's ring kernel) instead of a
# chain of whole-GPU elementwise kernels: 32 / 64 channels, with and without 20 us of link time per step.
o=gpurun_out/r5v; mkdir -p $o
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
python bench.py $B --force-allreduce 2>/dev/null | line "force_allreduce" | tee -a $o/ring_persistent.txt
SAE_RING_REHEARSAL_KERNEL=persistent:32:0 python bench.py $B --force-allreduce --ring-rehearsal 8 2>$o/err1.txt | line "ring8, one persistent kernel per bucket, 32 channels" | tee -a $o/ring_persistent.txt
SAE_RING_REHEARSAL_KERNEL=persistent:32:20 python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "ring8, persistent, 32 channels, 20 us per step (0.28 ms per bucket)" | tee -a $o/ring_persistent.txt
SAE_RING_REHEARSAL_KERNEL=persistent:64:20 python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "ring8, persistent, 64 channels, 20 us per step" | tee -a $o/ring_persistent.txt
SAE_RING_REHEARSAL_KERNEL=persistent:32:100 python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "ring8, persistent, 32 channels, 100 us per step (1.4 ms per bucket)" | tee -a $o/ring_persistent.txt
SAE_TWO_STREAMS=0 SAE_RING_REHEARSAL_KERNEL=persistent:32:20 python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "one stream: ring8, persistent, 32 channels, 20 us per step" | tee -a $o/ring_persistent.txt
tail -n 3 $o/err1.txt
echo SESSION_V_DONE

#!/bin/bash
# Round 5, GPU session N: is the ring rehearsal's cost (session M: +20 ms per step) contention for CUs / HBM, or streams sharing a
# hardware queue?  The same runs with 16 hardware queues, and with the step on one stream.
o=gpurun_out/r5n; mkdir -p $o
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
GPU_MAX_HW_QUEUES=16 python bench.py $B --force-allreduce 2>/dev/null | line "16_queues force_allreduce" | tee -a $o/ring_queues.txt
GPU_MAX_HW_QUEUES=16 python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "16_queues ring_rehearsal_8" | tee -a $o/ring_queues.txt
GPU_MAX_HW_QUEUES=24 python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "24_queues ring_rehearsal_8" | tee -a $o/ring_queues.txt
SAE_TWO_STREAMS=0 python bench.py $B --force-allreduce 2>/dev/null | line "one_stream force_allreduce" | tee -a $o/ring_queues.txt
SAE_TWO_STREAMS=0 python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "one_stream ring_rehearsal_8" | tee -a $o/ring_queues.txt
python bench.py $B --force-allreduce --ring-rehearsal 2 2>/dev/null | line "8_queues ring_rehearsal_2" | tee -a $o/ring_queues.txt
echo SESSION_N_DONE

#!/bin/bash
# Round 5, GPU session O: what about the ring rehearsal takes the two-stream overlap away -- variants of where its kernels run and of
# the priority of the step's side stream.
o=gpurun_out/r5o; mkdir -p $o
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
SAE_RING_REHEARSAL_STREAM=launch python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "ring8 on the launch stream" | tee -a $o/ring_variants.txt
SAE_RING_REHEARSAL_STREAM=high python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "ring8 on a high-priority stream" | tee -a $o/ring_variants.txt
SAE_SIDE_STREAM_PRIORITY=-1 python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "ring8, step side stream high priority" | tee -a $o/ring_variants.txt
SAE_SIDE_STREAM_PRIORITY=-1 python bench.py $B 2>/dev/null | line "plain, step side stream high priority" | tee -a $o/ring_variants.txt
SAE_SIDE_STREAM_PRIORITY=-1 python bench.py $B --force-allreduce 2>/dev/null | line "force_allreduce, step side stream high priority" | tee -a $o/ring_variants.txt
python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "ring8 (own stream, as session M)" | tee -a $o/ring_variants.txt
echo SESSION_O_DONE

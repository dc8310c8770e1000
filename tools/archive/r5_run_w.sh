#!/bin/bash
# Round 5, GPU session W: does the ring rehearsal's cost on two streams follow the NUMBER of collectives?  Bucket sizes 32 / 128 / 512 MB.
o=gpurun_out/r5w; mkdir -p $o
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'], '|', l['config'].get('force_allreduce','')[:90])"; }
export SAE_RING_REHEARSAL_KERNEL=persistent:32:20
for mb in 128 512; do
SAE_ALLREDUCE_BUCKET_MB=$mb python bench.py $B --force-allreduce 2>/dev/null | line "buckets of $mb MB: force_allreduce" | tee -a $o/ring_buckets.txt
SAE_ALLREDUCE_BUCKET_MB=$mb python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "buckets of $mb MB: ring8 persistent" | tee -a $o/ring_buckets.txt
done
SAE_ALLREDUCE_BUCKET_MB=512 SAE_TWO_STREAMS=0 python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "buckets of 512 MB, one stream: ring8 persistent" | tee -a $o/ring_buckets.txt
echo SESSION_W_DONE

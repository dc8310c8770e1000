#!/bin/bash
# Round 5, GPU session R: the deferred launch policy of the gradient all-reduce under the ring rehearsal (two-stream step), against
# the early one; the 1-rank RCCL tests.
o=gpurun_out/r5r; mkdir -p $o
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
for i in 1 2; do
python bench.py $B --force-allreduce 2>/dev/null | line "force_allreduce, deferred (default on two streams)" | tee -a $o/ring_policy.txt
python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "ring8 own stream, deferred (default)" | tee -a $o/ring_policy.txt
SAE_ALLREDUCE_LAUNCH=early python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "ring8 own stream, early" | tee -a $o/ring_policy.txt
done
SAE_TWO_STREAMS=0 python bench.py $B --force-allreduce --ring-rehearsal 8 2>/dev/null | line "one stream (early by default), ring8 own stream" | tee -a $o/ring_policy.txt
timeout 600 python -m pytest tests/test_gpu_allreduce.py tests/test_ddp_fullmodel.py -m gpu -q 2>&1 | tail -n 3 | tee $o/gpu_allreduce_tests.txt
echo SESSION_R_DONE

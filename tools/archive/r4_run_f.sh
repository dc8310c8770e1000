#!/bin/bash
# round-4 GPU session F: weight gradients on their own stream: values, then the step A/B
o=gpurun_out/r4F; mkdir -p $o
python -m pytest tests/test_gpu_determinism.py tests/test_gpu_parity.py tests/test_ddp_fullmodel.py tests/test_gpu_allreduce.py -m gpu -q -x > $o/gputests.log 2>&1; tail -4 $o/gputests.log
for aw in 0 1 0 1; do
  SAE_ASYNC_WGRAD=$aw python bench.py --steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing > $o/bench_aw$aw.json 2> $o/bench_aw$aw.err
  python - <<PY
import json
l = json.load(open("$o/bench_aw$aw.json"))
print("async_wgrad=$aw", l["value"], l["ms_per_step"], l["ms_d_call_median"], l["ms_g_call_median"], l["ms_r1_extra_max"])
PY
done
echo DONE

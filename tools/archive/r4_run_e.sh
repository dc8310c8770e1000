#!/bin/bash
# round-4 GPU session E: two HIP streams for the independent branches of the step: values, then the step A/B
o=gpurun_out/r4E; mkdir -p $o
python -m pytest tests/test_gpu_determinism.py tests/test_gpu_parity.py -m gpu -q -x > $o/gputests.log 2>&1; tail -4 $o/gputests.log
for ts in 0 1 0 1; do
  SAE_TWO_STREAMS=$ts python bench.py --steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing > $o/bench_ts$ts.json 2> $o/bench_ts$ts.err
  python - <<PY
import json
l = json.load(open("$o/bench_ts$ts.json"))
print("two_streams=$ts", l["value"], l["ms_per_step"], l["ms_d_call_median"], l["ms_g_call_median"], l["ms_r1_extra_max"])
PY
done
echo DONE

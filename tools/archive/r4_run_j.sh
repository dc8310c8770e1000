#!/bin/bash
# round-4 GPU session J: D(real) ahead of the generator passes on a third stream
o=gpurun_out/r4J; mkdir -p $o
for ed in 0 1 0 1; do
  SAE_EARLY_D_REAL=$ed python bench.py --steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing > $o/bench_ed$ed.json 2> $o/bench_ed$ed.err
  python - <<PY
import json
l = json.load(open("$o/bench_ed$ed.json"))
print("early_d_real=$ed", l["value"], l["ms_per_step"], l["ms_d_call_median"], l["ms_g_call_median"], l["ms_r1_extra_max"])
PY
done
echo DONE

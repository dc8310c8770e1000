#!/bin/bash
# round-4 GPU session H: prepared conv weights: tests, then the step A/B and the wprep launch count
o=gpurun_out/r4H; mkdir -p $o
python -m pytest tests/test_weight_prep.py tests/test_gpu_determinism.py tests/test_gpu_parity.py tests/test_modconv.py tests/test_styled_fused.py tests/test_resblock_fused.py tests/test_ddp_fullmodel.py -m gpu -q -x > $o/gputests.log 2>&1; tail -4 $o/gputests.log
for wc in 0 1 0 1; do
  SAE_WPREP_CACHE=$wc python bench.py --steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing > $o/bench_wc$wc.json 2> $o/bench_wc$wc.err
  python - <<PY
import json
l = json.load(open("$o/bench_wc$wc.json"))
print("wprep_cache=$wc", l["value"], l["ms_per_step"], l["ms_d_call_median"], l["ms_g_call_median"], l["ms_r1_extra_max"])
PY
done
export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats -d $root/$o/prof -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing > $root/$o/prof.log 2>&1
cd $root; python tools/prof_summary.py $o/prof 60 > $o/step_kernel_trace.txt 2>&1; grep -n "wprep\|dispatches" $o/step_kernel_trace.txt
find $o/prof -name "*.db" -delete; find $o/prof -name "*.csv" -size +1M -delete
echo DONE

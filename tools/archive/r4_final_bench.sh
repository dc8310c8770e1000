#!/bin/bash
# round-4: the driver's bench command on the final tree + the kernel traces the bench line is checked against
o=gpurun_out/r4_finalbench; mkdir -p $o
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; cut -c1-260 $o/bench_default.json
export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats -d $root/$o/prof -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing > $root/$o/prof.log 2>&1
cd $root; python tools/prof_summary.py $o/prof 45 > $o/step_church256_b16_f32_kernel_trace.txt 2>&1; head -6 $o/step_church256_b16_f32_kernel_trace.txt
find $o/prof -name "*.db" -delete; find $o/prof -name "*.csv" -size +1M -delete
cd /tmp
SAE_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats -d $root/$o/prof1 -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing > $root/$o/prof1.log 2>&1
cd $root; python tools/prof_summary.py $o/prof1 45 > $o/step_church256_b16_f32_kernel_trace_one_stream.txt 2>&1; head -8 $o/step_church256_b16_f32_kernel_trace_one_stream.txt
find $o/prof1 -name "*.db" -delete; find $o/prof1 -name "*.csv" -size +1M -delete
python - <<'PY'
import json
l = json.load(open("gpurun_out/r4_finalbench/bench_default.json"))
print(l["value"], l["ms_per_step"], l["frac_of_mfma_f32_roofline"], l["ms_per_step_one_stream"], l["roofline"]["frac"], l["roofline"]["avg_launch_ms"], l["roofline"]["launches"])
print([(k["class"][:22], k["ms_per_step"], k["frac"]) for k in l["roofline_by_kernel"]])
print("via_dropin", l["via_dropin"].get("dropin_over_direct"), "alt", l["alt_conv_math"]["value"])
PY
echo DONE

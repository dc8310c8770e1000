#!/bin/bash
# round-4 GPU session G: kernel trace of the step with two streams (resident / overlapped time), ledger by kernel
o=gpurun_out/r4G; mkdir -p $o
export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats -d $root/$o/prof -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing > $root/$o/prof.log 2>&1
cd $root; python tools/prof_summary.py $o/prof 45 > $o/step_church256_b16_f32_kernel_trace.txt 2>&1; head -30 $o/step_church256_b16_f32_kernel_trace.txt
find $o/prof -name "*.db" -delete; find $o/prof -name "*.csv" -size +1M -delete
tail -2 $o/prof.log | cut -c1-300
echo DONE

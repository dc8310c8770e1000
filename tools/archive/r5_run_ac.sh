#!/bin/bash
# Round 5, GPU session AC: rocprofv3 kernel trace of the bench command with the step on ONE stream (the mode bench.py's kernel pass --
# roofline.avg_launch_ms -- measures in), next to the two-stream trace of session K.
o=gpurun_out/r5ac; mkdir -p $o
root=$(pwd); export TMPDIR=/tmp; cd /tmp
SAE_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/trace -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets= > $root/$o/trace.log 2>&1
cd $root
f=$(find $o/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -60 "$f" > $o/kernel_stats_top60.csv
find $o -name "*.csv" -size +2M -delete; find $o -name "*.db" -delete
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $o/trace.log | head -n 2
head -n 5 $o/kernel_stats_top60.csv | cut -c1-200
echo SESSION_AC_DONE

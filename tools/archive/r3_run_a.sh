#!/bin/bash
# round-3 GPU session A: the new GPU tests, the allreduce rehearsal numbers + overlap trace, the PMC passes
o=gpurun_out/r3A; mkdir -p $o
python -m pytest tests/test_gpu_allreduce.py tests/test_dropin_standin.py tests/test_gpu_network_parity.py tests/test_quad_paths.py tests/test_f8_gather.py tests/test_resblock_fused.py -m gpu -q > $o/gputests.log 2>&1; tail -12 $o/gputests.log
cp gpurun_out/network_parity.jsonl $o/ 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 > $o/bench_plain.json 2> $o/bench_plain.err; cut -c1-300 $o/bench_plain.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 --force-allreduce > $o/bench_force_allreduce.json 2> $o/bench_force.err; cut -c1-300 $o/bench_force_allreduce.json; tail -3 $o/bench_force.err
export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $root/$o/ar_trace -- python $root/bench.py --steps 2 --warmup 2 --no-cpu-baseline --alt-steps 0 --no-kernel-timing --force-allreduce > $root/$o/ar_trace.log 2>&1
cd $root; python tools/allreduce_overlap.py $o/ar_trace > $o/allreduce_1rank_trace.txt 2>&1; cat $o/allreduce_1rank_trace.txt
find $o/ar_trace -name "*.csv" -size +1M -delete; find $o/ar_trace -name "*.db" -delete
bash tools/run_pmc.sh $o/pmc f32 > $o/pmc.log 2>&1; cat $o/pmc/pmc_dominant.json
echo DONE

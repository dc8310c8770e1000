#!/bin/bash
# Round-4 measurement run on the GPU box: everything DESIGN.md / profiles/ quote.  bash tools/r4_final_run.sh [tag]
t=${1:-final}; o=gpurun_out/r4_$t; mkdir -p $o
python -m pytest tests -m gpu -q > $o/gputests.log 2>&1; tail -3 $o/gputests.log
cp gpurun_out/network_parity.jsonl $o/ 2>/dev/null; cp gpurun_out/fullsize_parity.jsonl $o/ 2>/dev/null; cp gpurun_out/ddp_fullmodel_one_gpu.txt $o/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; cut -c1-300 $o/bench_default.json
python bench.py --conv-math bf16x6 --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 > $o/bench_bf16x6.json 2>/dev/null; cut -c1-200 $o/bench_bf16x6.json
python bench.py --preset tiny32 --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 > $o/bench_tiny32.json 2>/dev/null; cut -c1-200 $o/bench_tiny32.json
python bench.py --preset ffhq512 --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 > $o/bench_ffhq512.json 2>/dev/null; cut -c1-200 $o/bench_ffhq512.json
python bench.py --preset ffhq1024 --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 > $o/bench_ffhq1024.json 2>/dev/null; cut -c1-200 $o/bench_ffhq1024.json
python bench.py --steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --force-allreduce > $o/bench_force_allreduce.json 2> $o/bench_force_allreduce.err; cut -c1-200 $o/bench_force_allreduce.json
SAE_TWO_STREAMS=0 python tools/roofline_ledger.py --preset church256 --with-r1 --steps 16 2>&1 | grep -v amdgpu > $o/roofline_by_kernel_church256.txt
SAE_TWO_STREAMS=0 python tools/roofline_ledger.py --preset church256 --by-shape 2>&1 | grep -v amdgpu > $o/roofline_by_shape_church256.txt
SAE_TWO_STREAMS=0 python tools/roofline_ledger.py --preset ffhq512 2>&1 | grep -v amdgpu > $o/roofline_by_kernel_ffhq512.txt
SAE_TWO_STREAMS=0 python tools/roofline_ledger.py --preset ffhq1024 2>&1 | grep -v amdgpu > $o/roofline_by_kernel_ffhq1024.txt
tail -4 $o/roofline_by_kernel_*.txt | cut -c1-200
python tools/step_parity.py gpu > $o/step_parity_gpu.json 2>/dev/null; cut -c1-300 $o/step_parity_gpu.json
export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats -d $root/$o/prof -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing > $root/$o/prof.log 2>&1
cd $root; python tools/prof_summary.py $o/prof 45 > $o/step_church256_b16_f32_kernel_trace.txt 2>&1; head -8 $o/step_church256_b16_f32_kernel_trace.txt
find $o/prof -name "*.db" -delete; find $o/prof -name "*.csv" -size +1M -delete
cd /tmp
SAE_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats -d $root/$o/prof1 -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --no-kernel-timing > $root/$o/prof1.log 2>&1
cd $root; python tools/prof_summary.py $o/prof1 45 > $o/step_church256_b16_f32_kernel_trace_one_stream.txt 2>&1; head -8 $o/step_church256_b16_f32_kernel_trace_one_stream.txt
find $o/prof1 -name "*.db" -delete; find $o/prof1 -name "*.csv" -size +1M -delete
bash tools/run_pmc.sh $o/pmc f32 > $o/pmc.log 2>&1; cat $o/pmc/pmc_dominant.json | head -12
echo DONE

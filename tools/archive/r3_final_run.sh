#!/bin/bash
# Round-3 measurement run on the GPU box: everything DESIGN.md / profiles/ quote.  bash tools/r3_final_run.sh [tag]
t=${1:-final}; o=gpurun_out/r3_$t; mkdir -p $o
python -m pytest tests -m gpu -q > $o/gputests.log 2>&1; tail -3 $o/gputests.log
cp gpurun_out/network_parity.jsonl $o/ 2>/dev/null; cp gpurun_out/fullsize_parity.jsonl $o/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; cut -c1-300 $o/bench_default.json
python bench.py --conv-math bf16x6 --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 > $o/bench_bf16x6.json 2>/dev/null; cut -c1-200 $o/bench_bf16x6.json
python bench.py --preset tiny32 --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 > $o/bench_tiny32.json 2>/dev/null; cut -c1-200 $o/bench_tiny32.json
python bench.py --preset ffhq512 --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 > $o/bench_ffhq512.json 2>/dev/null; cut -c1-200 $o/bench_ffhq512.json
python bench.py --preset ffhq1024 --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 > $o/bench_ffhq1024.json 2>/dev/null; cut -c1-200 $o/bench_ffhq1024.json
python tools/roofline_ledger.py --preset church256 --with-r1 --steps 16 2>&1 | grep -v amdgpu > $o/roofline_by_kernel_church256.txt
python tools/roofline_ledger.py --preset church256 --by-shape 2>&1 | grep -v amdgpu > $o/roofline_by_shape_church256.txt
python tools/roofline_ledger.py --preset ffhq512 2>&1 | grep -v amdgpu > $o/roofline_by_kernel_ffhq512.txt
python tools/roofline_ledger.py --preset ffhq1024 2>&1 | grep -v amdgpu > $o/roofline_by_kernel_ffhq1024.txt
tail -4 $o/roofline_by_kernel_*.txt
python tools/step_parity.py gpu > $o/step_parity_gpu.json 2>/dev/null
python tools/torch_profile.py 2>&1 | grep -v amdgpu > $o/aten_glue_by_shape.txt; head -3 $o/aten_glue_by_shape.txt
python tools/aten_gpu_baseline.py 2>/dev/null | tail -1 > $o/stock_pytorch_rocm_discriminator.json; cat $o/stock_pytorch_rocm_discriminator.json
export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats -d $root/$o/prof -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --no-kernel-timing > $root/$o/prof.log 2>&1
cd $root; python tools/prof_summary.py $o/prof 45 > $o/step_church256_b16_f32_kernel_trace.txt 2>&1; head -14 $o/step_church256_b16_f32_kernel_trace.txt
find $o/prof -name "*.db" -delete; find $o/prof -name "*.csv" -size +1M -delete
bash tools/run_pmc.sh $o/pmc f32 > $o/pmc.log 2>&1; cat $o/pmc/pmc_dominant.json | head -12
bash tools/run_pmc_tr.sh $o/pmc_tr > $o/pmc_tr.log 2>&1; tail -4 $o/pmc_tr/summary.txt | cut -c1-400
echo DONE

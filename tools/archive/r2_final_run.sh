#!/bin/bash
# Round-2 measurement run on the GPU box: everything DESIGN.md / profiles/ quote.  bash tools/r2_final_run.sh [tag]
t=${1:-final}; o=gpurun_out; mkdir -p $o
python -m pytest tests -m gpu -q > $o/r2_gputests_$t.log 2>&1; tail -3 $o/r2_gputests_$t.log
python bench.py --steps 20 --warmup 5 > $o/r2_bench_default_$t.json 2> $o/r2_bench_default_$t.err; cat $o/r2_bench_default_$t.json
python bench.py --conv-math bf16x6 --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 > $o/r2_bench_bf16x6_$t.json 2>/dev/null; cut -c1-400 $o/r2_bench_bf16x6_$t.json
python bench.py --preset tiny32 --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 > $o/r2_bench_tiny32_$t.json 2>/dev/null; cut -c1-300 $o/r2_bench_tiny32_$t.json
python tools/roofline_ledger.py --preset church256 --with-r1 --steps 16 2>&1 | grep -v amdgpu > $o/r2_roofline_by_kernel_church256_$t.txt
python tools/roofline_ledger.py --preset ffhq512 2>&1 | grep -v amdgpu > $o/r2_roofline_by_kernel_ffhq512_$t.txt
python tools/roofline_ledger.py --preset ffhq1024 2>&1 | grep -v amdgpu > $o/r2_roofline_by_kernel_ffhq1024_$t.txt
python tools/roofline_ledger.py --preset church256 --conv-math bf16x6 2>&1 | grep -v amdgpu > $o/r2_roofline_by_kernel_church256_bf16x6_$t.txt
tail -4 $o/r2_roofline_by_kernel_*_$t.txt
python tools/step_parity.py gpu > $o/r2_step_parity_gpu_$t.json 2>/dev/null
python tools/torch_profile.py 2>&1 | grep -v amdgpu > $o/r2_aten_glue_$t.txt; head -3 $o/r2_aten_glue_$t.txt
(python tools/kb_subset.py; python tools/kb_1x1.py; python tools/kb_k1.py; python tools/kb_config3.py) 2>&1 | grep -v amdgpu > $o/r2_kernel_bench_$t.jsonl
export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats -d $root/$o/prof_r2_$t -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --no-kernel-timing > $root/$o/prof_r2_$t.log 2>&1
cd $root; python tools/prof_summary.py $o/prof_r2_$t 45 > $o/r2_step_church256_b16_f32_kernel_trace_$t.txt 2>&1; head -12 $o/r2_step_church256_b16_f32_kernel_trace_$t.txt
find $o/prof_r2_$t -name "*.db" -delete; find $o/prof_r2_$t -name "*.csv" -size +1M -delete
bash tools/run_pmc.sh $o/pmc_r2_f32_$t f32 > $o/pmc_r2_$t.log 2>&1
echo DONE

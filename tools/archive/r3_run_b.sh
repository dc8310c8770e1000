#!/bin/bash
# round-3 GPU session B: tests of the new paths, bench, ledger, PMC passes for the (new) dominant kernel
o=gpurun_out/r3B; mkdir -p $o
python -m pytest tests/test_gpu_allreduce.py tests/test_resblock_fused.py tests/test_gpu_network_parity.py tests/test_quad_paths.py -m gpu -q > $o/gputests.log 2>&1; tail -6 $o/gputests.log
cp gpurun_out/network_parity.jsonl $o/ 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 > $o/bench.json 2> $o/bench.err; cut -c1-250 $o/bench.json; tail -2 $o/bench.err
python tools/roofline_ledger.py --preset church256 --with-r1 --steps 16 2>&1 | grep -v amdgpu > $o/ledger.txt; head -22 $o/ledger.txt; tail -5 $o/ledger.txt
bash tools/run_pmc.sh $o/pmc f32 > $o/pmc.log 2>&1; cat $o/pmc/pmc_dominant.json
echo DONE

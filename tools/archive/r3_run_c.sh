#!/bin/bash
# round-3 GPU session C: kernel tests of the new entry points, step parity, bench, ledger, glue profile
o=gpurun_out/r3C; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py tests/test_resblock_fused.py tests/test_gpu_parity.py tests/test_gpu_determinism.py -m gpu -q -x > $o/gputests.log 2>&1; tail -5 $o/gputests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 > $o/bench.json 2> $o/bench.err; cut -c1-250 $o/bench.json; tail -2 $o/bench.err
python tools/roofline_ledger.py --preset church256 --with-r1 --steps 16 2>&1 | grep -v amdgpu > $o/ledger.txt; grep "hbm\|sum of\|outside\|wall" $o/ledger.txt | head -30
python tools/torch_profile.py 2>&1 | grep -v amdgpu > $o/aten_glue.txt; head -40 $o/aten_glue.txt
echo DONE

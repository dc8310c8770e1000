#!/bin/bash
# round-4 GPU session M: round-aware K split of mid-size gather launches; the bit-identity test that failed on a stale tuning build
o=gpurun_out/r4M; mkdir -p $o
python tools/ab_conv.py rsbase rsplit --op=fwd --op=dgrad --n24 --n40 --512@64 --512@32 2>&1 | tee $o/ab_round_split.txt | tail -16
python -m pytest tests/test_quad_paths.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_kernels.py -m gpu -q -x > $o/gputests.log 2>&1; tail -3 $o/gputests.log
echo DONE

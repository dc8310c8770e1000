#!/bin/bash
# round-4 GPU session B: wg16 (conv_wgrad16_kernel) variants against the round-3 weight-gradient kernels, isolated shapes
o=gpurun_out/r4B; mkdir -p $o
python tools/ab_conv.py "$@" --op=wgrad 2>&1 | tee $o/ab_wgrad_$1_$2.txt | tail -20
echo DONE

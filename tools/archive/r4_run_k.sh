#!/bin/bash
# round-4 GPU session K: roofline ledgers (one stream: the brackets need kernels that run alone)
o=gpurun_out/r4K; mkdir -p $o
export SAE_TWO_STREAMS=0
python tools/roofline_ledger.py --preset church256 --with-r1 --steps 16 2>&1 | grep -v amdgpu > $o/roofline_by_kernel_church256.txt
python tools/roofline_ledger.py --preset church256 --by-shape 2>&1 | grep -v amdgpu > $o/roofline_by_shape_church256.txt
head -75 $o/roofline_by_kernel_church256.txt | cut -c1-150
echo DONE

#!/bin/bash
# round-4 GPU session D: flat tiles of the transposed gather (tr2) on the small odd grids, isolated shapes
o=gpurun_out/r4D; mkdir -p $o
python tools/ab_conv.py "$@" --op=dgrad --s2 2>&1 | tee $o/ab_dgrad_s2.txt | tail -16
echo DONE

#!/bin/bash
# round-4 session N: the weight-gradient slice model and the gather's K-split price list on Dpatch's tail and the image
# discriminator's small maps (same box, one process per knob setting; the knobs are read once per process)
o=gpurun_out/r4_n; mkdir -p $o
F="--Dp3 --n24 --n40 --512@16 --512@32"
SAE_WGRAD_SLICE_MODEL=0 python tools/ab_conv.py tuning $F > $o/ab_base.txt 2>&1
python tools/ab_conv.py tuning $F > $o/ab_slices.txt 2>&1
SAE_IGEMM_SPLIT_MODEL=1 python tools/ab_conv.py tuning $F > $o/ab_split_model.txt 2>&1
paste -d'|' <(cut -c1-38 $o/ab_base.txt) <(cut -c28-38 $o/ab_slices.txt) <(cut -c28-38 $o/ab_split_model.txt)
echo DONE

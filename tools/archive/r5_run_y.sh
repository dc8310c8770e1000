#!/bin/bash
# Round 5, GPU session Y: counters of the stride-2 family at HEAD (gather, transposed gather, weight gradient on the step's three wide
# shapes): how busy the matrix pipe is, what the waves wait for, LDS conflicts, instruction mix.
o=gpurun_out/r5y; mkdir -p $o
root=$(pwd); export TMPDIR=/tmp; cd /tmp
W="$root/tools/pmc_kernels.py --s2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $root/$o/A -- python $W > $root/$o/A.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/$o/B -- python $W > $root/$o/B.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $root/$o/C -- python $W > $root/$o/C.log 2>&1
cd $root
python tools/pmc_summary.py $o/A $o/B $o/C > $o/pmc_summary.txt 2>&1
head -n 40 $o/pmc_summary.txt | cut -c1-250
find $o -name "*.csv" -size +2M -delete; find $o -name "*.db" -delete
echo SESSION_Y_DONE

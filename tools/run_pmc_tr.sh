#!/bin/bash
# counter passes over the transposed gather only (tools/pmc_kernels.py --tr): matrix-pipe busy + the request-level traffic
# counters.   bash tools/run_pmc_tr.sh gpurun_out/pmc_tr
out=${1:-gpurun_out/pmc_tr}
export TMPDIR=/tmp SAE_CONV_MATH=f32
root=$(pwd)
mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/$out/A -- python $root/tools/pmc_kernels.py --tr > $root/$out/A.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --kernel-trace --output-format csv -d $root/$out/E -- python $root/tools/pmc_kernels.py --tr > $root/$out/E.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum --kernel-trace --output-format csv -d $root/$out/F -- python $root/tools/pmc_kernels.py --tr > $root/$out/F.log 2>&1
cd $root
python tools/pmc_summary.py $out/A $out/E $out/F > $out/summary.txt 2>&1
find $out -name "*.csv" -size +2M -delete; find $out -name "*.db" -delete
cat $out/summary.txt

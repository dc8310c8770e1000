#!/bin/bash
# rocprofv3 counter passes over tools/pmc_kernels.py (each --pmc set in its own run, kernel trace only), merged by
# tools/pmc_summary.py.   bash tools/run_pmc.sh gpurun_out/pmc_r2 [f32|bf16x6]
out=${1:-gpurun_out/pmc}; math=${2:-f32}
export TMPDIR=/tmp SAE_CONV_MATH=$math
root=$(pwd)
mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $root/$out/A -- python $root/tools/pmc_kernels.py > $root/$out/A.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/$out/B -- python $root/tools/pmc_kernels.py > $root/$out/B.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/$out/C -- python $root/tools/pmc_kernels.py > $root/$out/C.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $root/$out/D -- python $root/tools/pmc_kernels.py > $root/$out/D.log 2>&1
# request-level view of the same traffic: FETCH_SIZE tallies 64 B per non-32B request although streaming reads are issued
# as 128 B requests (MI355X_MICROARCH.md, HBM); the raw request / miss counters bound the true figure per kernel
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --kernel-trace --output-format csv -d $root/$out/E -- python $root/tools/pmc_kernels.py > $root/$out/E.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum --kernel-trace --output-format csv -d $root/$out/F -- python $root/tools/pmc_kernels.py > $root/$out/F.log 2>&1
cd $root
python tools/pmc_summary.py $out/A $out/B $out/C $out/D $out/E $out/F --dominant-json $out/pmc_dominant.json > $out/summary.txt 2>&1
# keep only the summary and the logs (the raw csv / db files are large)
find $out -name "*.csv" -size +2M -delete; find $out -name "*.db" -delete
tail -60 $out/summary.txt

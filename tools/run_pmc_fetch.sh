#!/bin/bash
# quick pass: fabric read requests (x 128 B) and write requests of the hot kernels
out=${1:-gpurun_out/pmc_fetch}; root=$(pwd); mkdir -p $out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $root/$out/E -- python $root/tools/pmc_kernels.py > $root/$out/E.log 2>&1
cd $root; python tools/pmc_summary.py $out/E > $out/summary.txt 2>&1; find $out -name "*.csv" -size +2M -delete; find $out -name "*.db" -delete
grep -A1 "conv_igemm_kernel<3\|conv_igemm_kernel<1\|copyBuffer" $out/summary.txt | cut -c1-330

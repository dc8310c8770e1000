#!/bin/bash
# Round 5, GPU session F: the whole GPU suite at HEAD (new default routing), then traffic of the dominant kernel with the XCD-aware
# block order, the rocprofv3 kernel trace of the bench command, and the bench line.
o=gpurun_out/r5f; mkdir -p $o
rm -f gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $o/gpu_tests.log; tail -6 $o/gpu_tests.log
cp gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl $o/ 2>/dev/null
root=$(pwd); export TMPDIR=/tmp; cd /tmp
W=$root/tools/pmc_wino_fused.py
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $root/$o/A -- python $W > $root/$o/A.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/$o/B -- python $W > $root/$o/B.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --kernel-trace --output-format csv -d $root/$o/E -- python $W > $root/$o/E.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum --kernel-trace --output-format csv -d $root/$o/F -- python $W > $root/$o/F.log 2>&1
cd $root
python tools/pmc_summary.py $o/A $o/B $o/E $o/F --dominant-json $o/pmc_dominant.json > $o/pmc_summary.txt 2>&1
grep -A1 "^wino_fused" $o/pmc_summary.txt | cut -c1-200
grep -E "read_bytes|write_bytes|traffic_over" $o/pmc_dominant.json && cp $o/pmc_dominant.json profiles/r5_pmc_dominant.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/trace -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets= > $root/$o/trace.log 2>&1
cd $root
f=$(find $o/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $o/kernel_stats_top40.csv
find $o -name "*.csv" -size +2M -delete; find $o -name "*.db" -delete
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err || tail -5 $o/bench_default.err
python -c "
import json; l=json.loads(open('$o/bench_default.json').read().strip().splitlines()[-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'frac', l.get('frac_of_mfma_f32_roofline'))
print('roofline', {k:v for k,v in l.get('roofline',{}).items() if k!='note'})
for r in l.get('other_presets',[]): print('  preset', {k:r.get(k) for k in ('preset','value','ms_per_step','error')})
"
echo SESSION_F_DONE

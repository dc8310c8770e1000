#!/bin/bash
# Round 5, GPU session E: the one-kernel Winograd kernels with buffer-load staging (few vector-ALU instructions per chunk):
# parity, per-layer A/B, counters + HBM-side traffic of the dominant kernel (-> profiles/r5_pmc_dominant.json), the full bench line.
o=gpurun_out/r5e; mkdir -p $o
timeout 600 python -m pytest tests/test_winograd.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python tools/wino_ab.py --preset church256 > $o/wino_ab_church256.json 2> $o/wino_ab.err; tail -3 $o/wino_ab.err; tail -5 $o/wino_ab_church256.json
root=$(pwd); export TMPDIR=/tmp; cd /tmp
W=$root/tools/pmc_wino_fused.py
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $root/$o/A -- python $W > $root/$o/A.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/$o/B -- python $W > $root/$o/B.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --kernel-trace --output-format csv -d $root/$o/E -- python $W > $root/$o/E.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum --kernel-trace --output-format csv -d $root/$o/F -- python $W > $root/$o/F.log 2>&1
cd $root
python tools/pmc_summary.py $o/A $o/B $o/E $o/F --dominant-json $o/pmc_dominant.json > $o/pmc_summary.txt 2>&1
find $o -name "*.csv" -size +2M -delete; find $o -name "*.db" -delete
grep -A2 "^wino_fused" $o/pmc_summary.txt | cut -c1-420
cat $o/pmc_dominant.json && cp $o/pmc_dominant.json profiles/r5_pmc_dominant.json
timeout 900 python bench.py --steps 16 --warmup 4 > $o/bench_default.json 2> $o/bench_default.err || tail -5 $o/bench_default.err
python -c "
import json; l=json.loads(open('$o/bench_default.json').read().strip().splitlines()[-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'frac', l.get('frac_of_mfma_f32_roofline'))
print('roofline', {k:v for k,v in l.get('roofline',{}).items() if k!='note'})
for r in l.get('roofline_by_kernel',[]): print('  ', r['class'][:70], r['ms_per_step'], r['achieved'], r['frac'], r.get('frac_algorithmic'))
print('hbm', l.get('hbm_k1_k2'))
for r in l.get('other_presets',[]): print('  preset', {k:r.get(k) for k in ('preset','value','ms_per_step','frac_of_mfma_f32_roofline','error')})
print('cpu', l.get('cpu_baseline',{}).get('value'))
"
echo SESSION_E_DONE

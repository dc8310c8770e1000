#!/bin/bash
# Round 5, GPU session K: validation at HEAD -- smoke(), the whole GPU suite (with the full-size oracle cases of the one-kernel Winograd
# kernels), the per-kernel-class ledger of the step, the rocprofv3 kernel trace of the bench command, and the bench line.
o=gpurun_out/r5k; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $o/smoke.txt
rm -f gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $o/gpu_tests.log; tail -3 $o/gpu_tests.log
cp gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl $o/ 2>/dev/null
timeout 400 python tools/roofline_ledger.py --preset church256 --steps 16 > $o/roofline_by_kernel_church256.txt 2> $o/ledger.err || tail -3 $o/ledger.err
tail -4 $o/roofline_by_kernel_church256.txt
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err || tail -5 $o/bench_default.err
python -c "
import json
l=json.loads(open('$o/bench_default.json').read().strip().splitlines()[-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'frac', l.get('frac_of_mfma_f32_roofline'), 'one-stream', l.get('ms_per_step_one_stream'))
print('roofline', {k:v for k,v in l.get('roofline',{}).items() if k!='note'})
for r in l.get('roofline_by_kernel',[]): print('  ', r['class'][:70], r['ms_per_step'], r['achieved'], r['frac'], r.get('frac_algorithmic'))
print('hbm', l.get('hbm_k1_k2'))
for r in l.get('other_presets',[]): print('  preset', {k:r.get(k) for k in ('preset','value','ms_per_step','frac_of_mfma_f32_roofline','error')})
print('dropin', l.get('via_dropin',{}).get('dropin_over_direct'), 'cpu', l.get('cpu_baseline',{}).get('value'))
"
root=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/trace -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets= > $root/$o/trace.log 2>&1
cd $root
f=$(find $o/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -60 "$f" > $o/kernel_stats_top60.csv
find $o -name "*.csv" -size +2M -delete; find $o -name "*.db" -delete
echo SESSION_K_DONE

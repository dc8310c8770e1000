#!/bin/bash
# Round 5, GPU session T: final validation at HEAD -- smoke(), the whole GPU suite, the bench line, the ledgers of the two other presets.
o=gpurun_out/r5t; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 2 | tee $o/smoke.txt
rm -f gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 25 > $o/gpu_tests.log; tail -n 3 $o/gpu_tests.log
cp gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl $o/ 2>/dev/null
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err || tail -n 5 $o/bench_default.err
python -c "
import json
l=json.loads([x for x in open('$o/bench_default.json').read().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'frac', l.get('frac_of_mfma_f32_roofline'), 'one-stream', l.get('ms_per_step_one_stream'), l['config'].get('streams'))
print('roofline', {k:v for k,v in l.get('roofline',{}).items() if k!='note'})
print('hbm', l.get('hbm_k1_k2'))
for r in l.get('other_presets',[]): print('  preset', {k:r.get(k) for k in ('preset','value','ms_per_step','frac_of_mfma_f32_roofline','error')})
print('dropin', l.get('via_dropin',{}).get('dropin_over_direct'), 'cpu', l.get('cpu_baseline',{}).get('value'))
"
for p in ffhq512 ffhq1024; do
timeout 400 python tools/roofline_ledger.py --preset $p --steps 8 > $o/roofline_by_kernel_$p.txt 2> $o/ledger_$p.err || tail -n 3 $o/ledger_$p.err
tail -n 4 $o/roofline_by_kernel_$p.txt
done
echo SESSION_T_DONE

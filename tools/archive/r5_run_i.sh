#!/bin/bash
# Round 5, GPU session I: validation at HEAD -- smoke(), the whole GPU suite, the fused kernels' timings, the bench line (default and
# the single-rank rehearsal of the gradient all-reduce path).
o=gpurun_out/r5i; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
rm -f gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $o/gpu_tests.log; tail -3 $o/gpu_tests.log
cp gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl $o/ 2>/dev/null
python tools/wf_variants.py product 2>&1 | grep -v amdgpu | tee $o/wf_product.txt
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err || tail -5 $o/bench_default.err
timeout 600 python bench.py --force-allreduce --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets= > $o/bench_force_allreduce.json 2> $o/bench_force_allreduce.err || tail -5 $o/bench_force_allreduce.err
python -c "
import json
l=json.loads(open('$o/bench_default.json').read().strip().splitlines()[-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'frac', l.get('frac_of_mfma_f32_roofline'), 'one-stream', l.get('ms_per_step_one_stream'))
print('roofline', {k:v for k,v in l.get('roofline',{}).items() if k!='note'})
for r in l.get('roofline_by_kernel',[]): print('  ', r['class'][:70], r['ms_per_step'], r['achieved'], r['frac'], r.get('frac_algorithmic'))
print('hbm', l.get('hbm_k1_k2'))
for r in l.get('other_presets',[]): print('  preset', {k:r.get(k) for k in ('preset','value','ms_per_step','frac_of_mfma_f32_roofline','error')})
print('dropin', l.get('via_dropin',{}).get('dropin_over_direct'), 'cpu', l.get('cpu_baseline',{}).get('value'))
f=json.loads(open('$o/bench_force_allreduce.json').read().strip().splitlines()[-1]); print('force-allreduce', f['value'], f['ms_per_step'], f['config'].get('force_allreduce'))
"
echo SESSION_I_DONE

#!/bin/bash
# Round 6, GPU session T (after the streaming blur's noise prefetch): the GPU suite, the driver's bench command, the one-stream ledgers.
# (traces and PMC passes: session S, tools/archive/r6_run_s.sh -- the kernels they describe did not change)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6t; mkdir -p $o
rm -f gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl gpurun_out/step_parity_fullsize.jsonl
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -n 12 > $o/gpu_tests.log; tail -n 3 $o/gpu_tests.log | cut -c1-200
cp gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl gpurun_out/step_parity_fullsize.jsonl $o/ 2>/dev/null
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err || tail -5 $o/bench_default.err
python - <<'PY'
import json
l=json.loads([x for x in open('gpurun_out/r6t/bench_default.json').read().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'frac', l.get('frac_of_mfma_f32_roofline'), 'one-stream ms', l.get('ms_per_step_one_stream'), 'r1 extra', l.get('ms_r1_extra'))
print('roofline', {k:v for k,v in l.get('roofline',{}).items() if k!='note'})
print('hbm', l.get('hbm_k1_k2'))
print('other', [(r.get('preset'), r.get('value'), r.get('ms_per_step'), r.get('frac_of_mfma_f32_roofline')) for r in l.get('other_presets',[])])
print('alt', l.get('alt_conv_math')); print('dropin', {k:l.get('via_dropin',{}).get(k) for k in ('value','dropin_over_direct')})
print('cpu', l['cpu_baseline']['value'], l['cpu_baseline']['s_d_call'], l['cpu_baseline']['s_g_call'])
PY
for preset in church256 ffhq512 ffhq1024; do
python tools/roofline_ledger.py --preset $preset --steps 8 > $o/roofline_by_kernel_$preset.txt 2> $o/ledger_$preset.err || tail -3 $o/ledger_$preset.err
tail -n 4 $o/roofline_by_kernel_$preset.txt | cut -c1-200
done
python tools/roofline_ledger.py --preset church256 --steps 8 --by-shape > $o/roofline_by_shape_church256.txt 2>/dev/null
echo SESSION_T_DONE

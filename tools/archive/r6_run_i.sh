#!/bin/bash
# Round 6, GPU session I (final records): the whole GPU suite at HEAD, the driver's bench command, the one-stream ledgers of the
# three presets (per class and per conv geometry), the two-rank one-GPU rehearsal of the multi-rank line.
o=gpurun_out/r6i; mkdir -p $o
rm -f gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl gpurun_out/step_parity_fullsize.jsonl
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -n 12 > $o/gpu_tests.log; tail -n 4 $o/gpu_tests.log | cut -c1-200
cp gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl gpurun_out/step_parity_fullsize.jsonl $o/ 2>/dev/null
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err || tail -5 $o/bench_default.err
python - <<'PY'
import json
l=json.loads([x for x in open('gpurun_out/r6i/bench_default.json').read().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'frac', l.get('frac_of_mfma_f32_roofline'), 'one-stream ms', l.get('ms_per_step_one_stream'), 'r1 extra', l.get('ms_r1_extra'))
print('roofline', {k:v for k,v in l.get('roofline',{}).items() if k!='note'})
print('hbm', l.get('hbm_k1_k2'))
print('cpu', {k: l['cpu_baseline'].get(k) for k in ('value','cores','extrapolated','s_d_call','s_g_call')})
print('other', [(r.get('preset'), r.get('value'), r.get('ms_per_step'), r.get('frac_of_mfma_f32_roofline'), r.get('ms_r1_extra')) for r in l.get('other_presets',[])])
print('alt', l.get('alt_conv_math')); print('dropin', {k:l.get('via_dropin',{}).get(k) for k in ('value','dropin_over_direct')})
PY
for preset in church256 ffhq512 ffhq1024; do
python tools/roofline_ledger.py --preset $preset --steps 8 > $o/roofline_by_kernel_$preset.txt 2> $o/ledger_$preset.err || tail -3 $o/ledger_$preset.err
tail -n 4 $o/roofline_by_kernel_$preset.txt | cut -c1-200
done
python tools/roofline_ledger.py --preset church256 --steps 8 --by-shape > $o/roofline_by_shape_church256.txt 2>/dev/null; head -n 12 $o/roofline_by_shape_church256.txt | cut -c1-170
timeout 600 python bench.py --gpus 2 --same-device --steps 4 --warmup 2 --alt-steps 0 --kernel-steps 0 --no-kernel-timing --alt-streams-steps 2 --no-cpu-baseline > $o/two_ranks_church256.json 2> $o/two_ranks_church256.err; echo "two ranks staged rc=$?"
SAE_BENCH_SAME_DEVICE_STAGE=0 timeout 600 python bench.py --gpus 2 --same-device --steps 4 --warmup 2 --alt-steps 0 --kernel-steps 0 --no-kernel-timing --alt-streams-steps 2 --no-cpu-baseline > $o/two_ranks_church256_unstaged.json 2> $o/two_ranks_church256_unstaged.err; echo "two ranks unstaged rc=$?"
python - <<'PY'
import json
for n in ('two_ranks_church256','two_ranks_church256_unstaged'):
    try:
        l=json.loads([x for x in open('gpurun_out/r6i/%s.json'%n).read().splitlines() if x.startswith('{')][-1])
        print(n, l['value'], l['ms_per_step'], l.get('ms_per_step_by_rank'), {k:(v if not isinstance(v,dict) else {a:b for a,b in v.items()}) for k,v in l['allreduce'].items() if k!='note'}, l.get('alt_streams'))
    except Exception as e: print(n, 'no line', e)
PY
echo SESSION_I_DONE

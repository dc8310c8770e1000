#!/bin/bash
# Round 6, GPU session J: smoke(), the whole GPU suite (MIOpen immediate find for the controls), gemm rule A/B in the ledger, bench.
o=gpurun_out/r6j; mkdir -p $o
python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids" | tail -n 3 | tee $o/smoke.txt
rm -f gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl gpurun_out/step_parity_fullsize.jsonl
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | grep -v "amdgpu.ids" | tail -n 16 > $o/gpu_tests.log; tail -n 14 $o/gpu_tests.log | cut -c1-200
cp gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl gpurun_out/step_parity_fullsize.jsonl $o/ 2>/dev/null
python tools/roofline_ledger.py --preset church256 --steps 8 > $o/roofline_by_kernel_church256.txt 2>/dev/null; grep "gemm\|noise + bias + lrelu forward\|sum of\|outside\|wall per" $o/roofline_by_kernel_church256.txt | cut -c1-170
python tools/roofline_ledger.py --preset church256 --steps 8 --by-shape 2>/dev/null | grep "gemm (linear" | head -12 | cut -c1-150 | tee $o/gemm_by_shape.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err || tail -5 $o/bench_default.err
python - <<'PY'
import json
l=json.loads([x for x in open('gpurun_out/r6j/bench_default.json').read().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'frac', l.get('frac_of_mfma_f32_roofline'), 'r1 extra', l.get('ms_r1_extra'), 'roofline', l['roofline']['frac'], 'hbm', l['hbm_k1_k2']['achieved'])
print('other', [(r.get('preset'), r.get('value'), r.get('frac_of_mfma_f32_roofline'), r.get('ms_r1_extra')) for r in l.get('other_presets',[])], 'cpu', l['cpu_baseline']['value'], l['cpu_baseline']['extrapolated'])
PY
echo SESSION_J_DONE

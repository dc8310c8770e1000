#!/bin/bash
# Round 6, GPU session A: HEAD after the advisor fixes -- Winograd tests on the GPU, the default bench line (new: cpu_baseline
# measured on a whole B = 2 iteration), one-stream step for the launch-gap figure.
o=gpurun_out/r6a; mkdir -p $o
timeout 900 python -m pytest tests/test_winograd.py tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -n 3 | tee $o/gpu_subset.txt
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; tail -c 600 $o/bench_default.err
python - <<'PY'
import json
l=json.loads([x for x in open('gpurun_out/r6a/bench_default.json').read().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'roofline', l['roofline']['frac'], 'cpu', {k: l['cpu_baseline'].get(k) for k in ('value','cores','extrapolated','s_d_call','s_g_call')})
print('other', {k:(v.get('value'),v.get('ms_per_step')) for k,v in l.get('other_presets',{}).items()})
PY
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
SAE_TWO_STREAMS=0 python bench.py $B 2>/dev/null | tail -n 1 | cut -c1-400 | tee $o/bench_one_stream.json
echo SESSION_A_DONE

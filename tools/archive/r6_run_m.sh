#!/bin/bash
# round 6 session M: whole GPU suite + bench with the new schedules and the polyphase route
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6m
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r6m/gpu_tests.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r6m/bench.json 2> gpurun_out/r6m/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6m/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
for r in d.get('roofline_by_kernel',[]): print(r['class'][:60], r['ms_per_step'], r['frac'])
PY
echo SESSION_M_DONE

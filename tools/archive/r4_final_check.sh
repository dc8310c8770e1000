#!/bin/bash
# round-4 last session: the whole GPU suite and the driver's bench command on the final tree, then the measured CPU baseline
o=gpurun_out/r4_check; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python -m pytest tests -m gpu -q > $o/gputests.log 2>&1; tail -2 $o/gputests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; cut -c1-260 $o/bench_default.json
python bench.py --steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --force-allreduce 2>/dev/null | grep '^{' > $o/bench_force_allreduce.json; cut -c1-200 $o/bench_force_allreduce.json
python bench.py --steps 16 --warmup 3 --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --full-cpu-baseline > $o/bench_full_cpu_baseline.json 2> $o/bench_full_cpu.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r4_check/bench_full_cpu_baseline.json"))
print("CPU", json.dumps(l["cpu_baseline"])[:900])
PY
echo DONE

#!/bin/bash
# Round 6, GPU session C: the raw-buffer range check probed directly; Winograd / Adam tests at HEAD; the hipGraph step -- parity
# with the eager step, then the bench with and without graphs (two streams and one), tiny32 and ffhq1024.
o=gpurun_out/r6c; mkdir -p $o
(cd tools/probe && ./buffer_range_probe) 2>&1 | grep -v amdgpu.ids | tee $o/buffer_range_check.txt
timeout 600 python -m pytest tests/test_winograd.py tests/test_adam.py -m gpu -q 2>&1 | tail -n 5 | tee $o/wino_adam.txt
timeout 900 python -m pytest tests/test_gpu_graph_step.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -n 40 | tee $o/graph_tests.txt
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l.get('ms_d_call_median'), l.get('ms_g_call_median'), l['config'].get('launch','')[:40])"; }
for i in 1 2; do
python bench.py $B 2>$o/err_graph.txt | line graph+2streams | tee -a $o/step_ab.txt
python bench.py $B --no-graph 2>/dev/null | line eager+2streams | tee -a $o/step_ab.txt
done
SAE_TWO_STREAMS=0 python bench.py $B 2>/dev/null | line graph+1stream | tee -a $o/step_ab.txt
SAE_TWO_STREAMS=0 python bench.py $B --no-graph 2>/dev/null | line eager+1stream | tee -a $o/step_ab.txt
python bench.py $B --preset tiny32 2>/dev/null | line tiny32-graph | tee -a $o/step_ab.txt
python bench.py $B --preset tiny32 --no-graph 2>/dev/null | line tiny32-eager | tee -a $o/step_ab.txt
python bench.py $B --preset ffhq1024 2>/dev/null | line ffhq1024-graph | tee -a $o/step_ab.txt
python bench.py $B --preset ffhq1024 --no-graph 2>/dev/null | line ffhq1024-eager | tee -a $o/step_ab.txt
tail -n 5 $o/err_graph.txt
echo SESSION_C_DONE

#!/bin/bash
# Round 6, GPU session K: the generator's two passes as ONE batched pass (SAE_G_BATCHED=1, networks/generator.py: forward_pair) against
# two passes on two streams -- step A/B in graph mode (and eager one-stream, the multi-rank mode), parity of the batched form.
o=gpurun_out/r6k; mkdir -p $o
SAE_G_BATCHED=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph_step.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -n 4 | tee $o/parity_batched.txt
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l.get('ms_d_call_median'), l.get('ms_g_call_median'))"; }
for i in 1 2; do
python bench.py $B 2>/dev/null | line "church256 two passes (graph, 2 streams)" | tee -a $o/ab.txt
SAE_G_BATCHED=1 python bench.py $B 2>/dev/null | line "church256 one batched pass (graph, 2 streams)" | tee -a $o/ab.txt
done
SAE_TWO_STREAMS=0 python bench.py $B --no-graph 2>/dev/null | line "church256 two passes (eager, 1 stream)" | tee -a $o/ab.txt
SAE_G_BATCHED=1 SAE_TWO_STREAMS=0 python bench.py $B --no-graph 2>/dev/null | line "church256 one batched pass (eager, 1 stream)" | tee -a $o/ab.txt
python bench.py $B --preset ffhq512 2>/dev/null | line "ffhq512 two passes" | tee -a $o/ab.txt
SAE_G_BATCHED=1 python bench.py $B --preset ffhq512 2>/dev/null | line "ffhq512 one batched pass" | tee -a $o/ab.txt
python bench.py $B --preset ffhq1024 2>/dev/null | line "ffhq1024 two passes" | tee -a $o/ab.txt
SAE_G_BATCHED=1 python bench.py $B --preset ffhq1024 2>/dev/null | line "ffhq1024 one batched pass" | tee -a $o/ab.txt
echo SESSION_K_DONE

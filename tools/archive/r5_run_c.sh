#!/bin/bash
# Round 5, GPU session C: the one-kernel Winograd route (csrc/winograd_fused.hip) -- parity on the GPU, per-layer A/B against the direct
# kernels and the three-kernel route, and the step with: default routing / three-kernel form only / no Winograd.
o=gpurun_out/r5c; mkdir -p $o
timeout 600 python -m pytest tests/test_winograd.py tests/test_weight_prep.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python tools/wino_ab.py --preset church256 > $o/wino_ab_church256.json 2> $o/wino_ab.err; tail -3 $o/wino_ab.err; tail -5 $o/wino_ab_church256.json
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
python bench.py $B 2>$o/bench_default.err | line default-routing || tail -5 $o/bench_default.err
python bench.py $B --no-winograd-fused 2>/dev/null | line three-kernel-only
python bench.py $B --no-winograd 2>/dev/null | line direct-only
python bench.py $B 2>/dev/null | line default-routing
echo SESSION_C_DONE

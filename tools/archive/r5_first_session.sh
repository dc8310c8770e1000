#!/bin/bash
# First GPU session of the next round: the Winograd F(2x2,3x3) route -- parity on the GPU, then the step with and
# without it on ONE box (two rounds each), a channel-threshold sweep, and the per-shape ledger with the route on.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r5_first_session.sh'
o=gpurun_out/r5_wino; mkdir -p $o
python -m pytest tests/test_winograd.py tests/test_ws_gather.py -m gpu -x -q 2>&1 | tail -3
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing"
line() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
for r in 1 2; do
  python bench.py $B 2>/dev/null | line direct
  python bench.py $B --winograd 2>/dev/null | line winograd-256
done
SAE_WINOGRAD_MIN_C=512 python bench.py $B --winograd 2>/dev/null | line winograd-512
SAE_WINOGRAD_MIN_C=128 python bench.py $B --winograd 2>/dev/null | line winograd-128
# per-shape ledger with the route on (classes: "winograd products", "winograd ... transform")
SAE_WINOGRAD=1 python tools/roofline_ledger.py --by-shape --steps 4 > $o/roofline_by_shape_winograd.txt 2> $o/ledger.err || tail -3 $o/ledger.err
grep -i "winograd" $o/roofline_by_shape_winograd.txt | sort -k1,1 | head -40
echo DONE

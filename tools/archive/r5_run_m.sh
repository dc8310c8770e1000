#!/bin/bash
# Round 5, GPU session M: the step's ledger on one stream (per kernel class and per shape), and the ring all-reduce rehearsal
# (bench.py --force-allreduce --ring-rehearsal 8) against the plain forced path, same box, twice each.
o=gpurun_out/r5m; mkdir -p $o
if [ -z "$SKIP_LEDGER" ]; then
timeout 400 python tools/roofline_ledger.py --preset church256 --steps 16 > $o/roofline_by_kernel_church256.txt 2> $o/ledger.err || tail -n 3 $o/ledger.err
tail -n 4 $o/roofline_by_kernel_church256.txt
timeout 400 python tools/roofline_ledger.py --preset church256 --steps 16 --by-shape > $o/roofline_by_shape_church256.txt 2> $o/ledger2.err || tail -n 3 $o/ledger2.err
fi
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
for i in 1 2; do
python bench.py $B 2>/dev/null | line plain | tee -a $o/ring.txt
python bench.py $B --force-allreduce 2>/dev/null | line force_allreduce | tee -a $o/ring.txt
python bench.py $B --force-allreduce --ring-rehearsal 8 2>$o/ring.err | line ring_rehearsal_8 | tee -a $o/ring.txt
done
tail -n 3 $o/ring.err
echo SESSION_M_DONE

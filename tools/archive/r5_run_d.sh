#!/bin/bash
# Round 5, GPU session D: the one-kernel Winograd weight gradient -- parity, per-layer A/B, the step, counters of both fused kernels.
o=gpurun_out/r5d; mkdir -p $o
timeout 600 python -m pytest tests/test_winograd.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python tools/wino_ab.py --preset church256 > $o/wino_ab_church256.json 2> $o/wino_ab.err; tail -3 $o/wino_ab.err; tail -5 $o/wino_ab_church256.json
B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'])"; }
python bench.py $B 2>$o/bench_default.err | line default-routing || tail -5 $o/bench_default.err
python bench.py $B 2>/dev/null | line default-routing
root=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $root/$o/A -- python $root/tools/pmc_wino_fused.py > $root/$o/A.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/$o/B -- python $root/tools/pmc_wino_fused.py > $root/$o/B.log 2>&1
cd $root
python tools/pmc_summary.py $o/A $o/B > $o/pmc_summary.txt 2>&1
find $o -name "*.csv" -size +2M -delete; find $o -name "*.db" -delete
grep -A12 "wino_fused" $o/pmc_summary.txt | head -80
echo SESSION_D_DONE

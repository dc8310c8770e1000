#!/bin/bash
# Round 6, GPU session V (final records of the round): the driver's bench command, kernel traces of it (graph + two
# streams = the timed mode; eager + one stream = the kernel pass's mode), PMC passes of the dominant kernel, the one-stream ledgers.
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r6v; mkdir -p $o
rm -f gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl gpurun_out/step_parity_fullsize.jsonl
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -n 12 > $o/gpu_tests.log; tail -n 3 $o/gpu_tests.log | cut -c1-200
cp gpurun_out/network_parity*.jsonl gpurun_out/fullsize_parity.jsonl gpurun_out/step_parity_fullsize.jsonl $o/ 2>/dev/null
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err || tail -5 $o/bench_default.err
python - <<'PY'
import json
l=json.loads([x for x in open('gpurun_out/r6v/bench_default.json').read().splitlines() if x.startswith('{')][-1])
print('value', l['value'], 'ms', l['ms_per_step'], 'frac', l.get('frac_of_mfma_f32_roofline'), 'one-stream ms', l.get('ms_per_step_one_stream'), 'r1 extra', l.get('ms_r1_extra'))
print('roofline', {k:v for k,v in l.get('roofline',{}).items() if k!='note'})
print('hbm', l.get('hbm_k1_k2'))
print('other', [(r.get('preset'), r.get('value'), r.get('ms_per_step'), r.get('frac_of_mfma_f32_roofline')) for r in l.get('other_presets',[])])
print('alt', l.get('alt_conv_math')); print('dropin', {k:l.get('via_dropin',{}).get(k) for k in ('value','dropin_over_direct')})
PY
root=$(pwd); export TMPDIR=/tmp; cd /tmp
B="--steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/trace_graph -- python $root/bench.py $B > $root/$o/trace_graph.log 2>&1
SAE_TWO_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/trace_one -- python $root/bench.py $B --no-graph > $root/$o/trace_one.log 2>&1
cd $root
for t in trace_graph trace_one; do f=$(find $o/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -70 "$f" > $o/${t}_kernel_stats_top70.csv; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $o/$t.log | head -n 2; head -n 4 $o/${t}_kernel_stats_top70.csv | cut -c1-160; done
find $o -name "*.csv" -size +2M -delete; find $o -name "*.db" -delete
cd /tmp
W=$root/tools/pmc_wino_fused.py
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $root/$o/A -- python $W > $root/$o/A.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/$o/B -- python $W > $root/$o/B.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --kernel-trace --output-format csv -d $root/$o/E -- python $W > $root/$o/E.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum --kernel-trace --output-format csv -d $root/$o/F -- python $W > $root/$o/F.log 2>&1
cd $root
python tools/pmc_summary.py $o/A $o/B $o/E $o/F --dominant-json $o/pmc_dominant.json > $o/pmc_summary.txt 2>&1
grep -A1 "^wino_fused" $o/pmc_summary.txt | cut -c1-220
grep -E "read_bytes|write_bytes|traffic_over|kernel" $o/pmc_dominant.json | head
find $o -name "*.csv" -size +2M -delete; find $o -name "*.db" -delete
for preset in church256 ffhq512 ffhq1024; do
python tools/roofline_ledger.py --preset $preset --steps 8 > $o/roofline_by_kernel_$preset.txt 2> $o/ledger_$preset.err || tail -3 $o/ledger_$preset.err
tail -n 4 $o/roofline_by_kernel_$preset.txt | cut -c1-200
done
python tools/roofline_ledger.py --preset church256 --steps 8 --by-shape > $o/roofline_by_shape_church256.txt 2>/dev/null; head -n 12 $o/roofline_by_shape_church256.txt | cut -c1-170
timeout 600 python bench.py --gpus 2 --same-device --steps 4 --warmup 2 --alt-steps 0 --kernel-steps 0 --no-kernel-timing --alt-streams-steps 2 --no-cpu-baseline > $o/two_ranks_church256.json 2> $o/two_ranks_church256.err; echo "two ranks staged rc=$?"
tail -c 600 $o/two_ranks_church256.json
echo SESSION_V_DONE

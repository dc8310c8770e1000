#!/bin/bash
# Round 6, GPU session H: the lazy-R1 surcharge of ffhq512 in graph mode (336 ms in session G against 57 in round 5), counters of the
# K1 blur kernels, K1 A/B after the routing rule.
o=gpurun_out/r6h; mkdir -p $o
B="--no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l.get('ms_d_call_median'), l.get('ms_g_call_median'), 'r1 extra', l.get('ms_r1_extra'), 'r1 in window', l.get('r1_iterations_in_window'))"; }
python bench.py --preset ffhq512 --steps 16 --warmup 3 $B 2>/dev/null | line "ffhq512 graph 16" | tee -a $o/ffhq512_r1.txt
python bench.py --preset ffhq512 --steps 32 --warmup 3 $B 2>/dev/null | line "ffhq512 graph 32" | tee -a $o/ffhq512_r1.txt
python bench.py --preset ffhq512 --steps 16 --warmup 3 $B --no-graph 2>/dev/null | line "ffhq512 eager 16" | tee -a $o/ffhq512_r1.txt
python bench.py --preset ffhq512 --steps 32 --warmup 3 $B --no-graph 2>/dev/null | line "ffhq512 eager 32" | tee -a $o/ffhq512_r1.txt
python bench.py --preset church256 --steps 32 --warmup 3 $B 2>/dev/null | line "church256 graph 32" | tee -a $o/ffhq512_r1.txt
root=$(pwd); export TMPDIR=/tmp; cd /tmp
W=$root/tools/pmc_k1.py
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/$o/A -- python $W > $root/$o/A.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --kernel-trace --output-format csv -d $root/$o/B -- python $W > $root/$o/B.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --kernel-trace --output-format csv -d $root/$o/E -- python $W > $root/$o/E.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum --kernel-trace --output-format csv -d $root/$o/F -- python $W > $root/$o/F.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $root/$o/G -- python $W > $root/$o/G.log 2>&1
cd $root
python tools/pmc_summary.py $o/A $o/B $o/E $o/F $o/G > $o/pmc_k1_summary.txt 2>&1
grep -A1 "^blur\|^void sae.*blur\|blur_" $o/pmc_k1_summary.txt | cut -c1-900 | head -40
tail -n 3 $o/A.log $o/G.log | cut -c1-200
find $o -name "*.csv" -size +2M -delete; find $o -name "*.db" -delete
echo SESSION_H_DONE
